"""Backward of ``CustomCondUNet2DModel`` on MI355X: what ``accelerator.backward(loss)`` (``utils_training.py:436``) asks
autograd to do over the reference's UNet, as a static plan of hand-written HIP launches.

``UNetTrainPlan`` lays out the forward exactly as :class:`phendiff_amd.unet.UNetPlan` does (same kernels, every activation
already lives in its own buffer) and additionally keeps the GroupNorm statistics and the attention log-sum-exp; it then
walks the forward's block tape in reverse and emits, per block:

* convolution input gradients: ``pd_conv`` with transposed + flipped weights (zero-stuffed for the stride-2 convs, pooled
  2x2 for the convs fused with the nearest upsample); gradient accumulation over skip connections rides in ``residual``;
* weight gradients: ``pd_conv_wgrad`` (pixel-K MFMA GEMM, the GroupNorm-affine + SiLU input rebuilt on the fly);
* GroupNorm(+SiLU) backward: ``pd_gn_silu_bwd`` (also folds the identity-skip / shortcut gradient into ``dx``);
* attention backward: ``pd_attn_d8_bwd``; bias / time-embedding-projection gradients: ``pd_channel_sum``;
* the fp32 time-embedding path: ``pd_linear_wgrad`` / ``pd_linear_dgrad`` / ``pd_embedding_grad``.

Parameter gradients are accumulated (``+=``) in fp32 into caller-provided tensors that alias one flat buffer
(:func:`training_param_order` gives the parameter order that makes the fused q/k/v and stacked ``time_emb_proj``
gradients contiguous), which is what :class:`phendiff_amd.training.FlatAdamWEMA` updates in a single pass.
"""
from __future__ import annotations

import ctypes as C
import os
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from .packing import dgrad_weight, pack_conv_weight, upsample_phase_weights_stacked
from .unet import CustomCondUNet2DModel, UNetPlan, _Attention, _Op, _PackedWeights, _Resnet, _Sampler, _copy_into


# diagnostic (same-box A/B): route the 1x1 gradients through pd_conv / pd_conv_wgrad as the 3x3 ones
_NO_LINEAR_GRADS = bool(os.environ.get("PD_NO_LINEAR_GRADS"))
# Round 6: every weight-gradient launch is two kernels -- the GEMM that leaves per-split partial tiles in a slab, and the ordered fold of the slab
# into the fp32 gradient (bandwidth-bound, few workgroups, ~150 + 70 of them per SD-2.1 step).  With ONE slab PER LAUNCH (288 GB of HBM: ~18 GB for the
# SD-2.1 UNet at B = 32) the folds run on a second stream under the next layers' GEMMs.  PD_WGRAD_SIDE=0: one shared slab, one stream (same-box A/B).
_WGRAD_SIDE = os.environ.get("PD_WGRAD_SIDE", "1") != "0"
# diagnostic (same-box A/B): keep the GroupNorm-prologue 1x1 weight gradients on pd_conv_wgrad (round 3 routes them through
# pd_gn_apply + pd_token_wgrad)
_NO_PREAPPLY_WGRAD = bool(os.environ.get("PD_NO_PREAPPLY_WGRAD"))

def training_param_order(m: CustomCondUNet2DModel) -> List[Tuple[str, torch.nn.Parameter]]:
    """(name, parameter) pairs in the order the flat training buffers use: all ``time_emb_proj`` weights (then biases)
    stacked in module order -- one [proj_dim][tdim] matrix, as ``pd_temb`` sees them --, each attention's
    to_q / to_k / to_v weights (then biases) adjacent -- the fused [3C][C] projection --, then everything else, the class
    embedding's parameters LAST (they receive no gradient on unconditional steps: torch's AdamW then skips them, so they are a tail
    segment of the flat optimizer with its own step count)."""
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
        if isinstance(mod, _Attention):
            for suffix in ("weight", "bias"):
                for which in ("to_q", "to_k", "to_v"):
                    take(f"{n}.{which}.{suffix}")
    for n in named:
        if n not in seen and not n.startswith("class_embedding."):
            take(n)
    # the class embedding LAST: the nn.Embedding table, or (class_embed_type = "timestep", cond_unet_2d.py:146-153) the four tensors
    # of the class MLP -- none of them receives a gradient on an unconditional step
    for n in named:
        if n.startswith("class_embedding."):
            take(n)
    return out


def _contiguous_after(a: torch.Tensor, b: torch.Tensor) -> bool:
    return b.data_ptr() == a.data_ptr() + a.numel() * a.element_size()


class TrainWeights:
    """Input-gradient ("dgrad") weights in ``pd_conv``'s packed layout: W'[ci][co][ky][kx] = W[co][ci][K-1-ky][K-1-kx]."""

    def __init__(self, m: CustomCondUNet2DModel, device, tdt):
        # fp16 (round 5): the reference's `--mixed_precision fp16` training (launch_script_DDIM.sh:56) -- fp16 activations and activation
        # gradients on the f16 MFMA, fp32 master weights / parameter gradients, a loss scale (training.LossScaler) set up by UNetTrainer
        self.tdt, self.device = tdt, device
        pk = lambda w: pack_conv_weight(dgrad_weight(w.detach().to(device=device, dtype=torch.float32)), tdt)
        lin = lambda w: w.detach()[:, :, None, None]
        self.resnets, self.attns, self.samplers = {}, {}, {}
        for name, mod in m.named_modules():
            if isinstance(mod, _Resnet):
                e = SimpleNamespace(w1d=pk(mod.conv1.weight), w2d=pk(mod.conv2.weight))
                if mod.conv_shortcut is not None:
                    e.wsd = pk(mod.conv_shortcut.weight)
                self.resnets[name] = e
            elif isinstance(mod, _Attention):
                wqkv = torch.cat([mod.to_q.weight, mod.to_k.weight, mod.to_v.weight], 0)
                self.attns[name] = SimpleNamespace(wqkvd=pk(lin(wqkv)), wod=pk(lin(mod.to_out[0].weight)))
            elif isinstance(mod, _Sampler):
                self.samplers[name] = SimpleNamespace(wd=pk(mod.conv.weight))
                if ".upsamplers." in name:      # input-gradient weights of the four sub-pixel phases (pd_conv phase_in): transposed, taps flipped
                    k4 = upsample_phase_weights_stacked(mod.conv.weight.detach().to(device=device, dtype=torch.float32))
                    self.samplers[name].wd4 = tuple(pk(k4[p]) for p in range(4))
        co = m.conv_out.weight.shape[0]
        wo = torch.zeros((((co + 31) // 32) * 32,) + tuple(m.conv_out.weight.shape[1:]), dtype=torch.float32, device=device)
        wo[:co] = m.conv_out.weight.detach().to(device=device, dtype=torch.float32)
        self.conv_out_d = pk(wo)           # 32 (3 real) output-gradient channels -> block_out_channels[0]
        # conv_in's input gradient (the image gradient of the guided transfer): block_out_channels[0] -> 3 channels (pad 32)
        self.conv_in_d = pack_conv_weight(dgrad_weight(m.conv_in.weight.detach().to(device=device, dtype=torch.float32)), tdt, 32)

    def refresh(self, m):
        _copy_into(self, TrainWeights(m, self.device, self.tdt))


class UNetTrainPlan(UNetPlan):
    """Forward (with saved statistics) + backward launch plan for a fixed (B, H, W).

    ``params`` / ``grads``: state_dict-name -> fp32 device tensor (master parameter / its gradient), laid out as
    :func:`training_param_order` prescribes.  ``backward`` ACCUMULATES into ``grads`` (zero them between steps)."""

    # Upsample2D runs as four sub-pixel 2x2 phases in the training forward as well (round 4): its input gradient is the four phases run
    # backwards (pd_conv phase_in: 4 / 9 of the FLOPs of the 3x3 input gradient over the upsampled tensor, and no 2x2 pooling pass);
    # the weight gradient differentiates the 3x3 weights over the nearest-upsampled input as before
    subpixel_in_training = True

    def __init__(self, m: CustomCondUNet2DModel, w: _PackedWeights, tw: TrainWeights, B, H, W, device,
                 params: Optional[Dict[str, torch.Tensor]] = None, grads: Optional[Dict[str, torch.Tensor]] = None,
                 input_grad: bool = False, frozen=()):
        """``grads`` = None: no parameter gradients (weight / bias / GroupNorm / embedding launches are not emitted);
        ``input_grad``: also produce d loss / d sample (fp32 NCHW, ``self.dsample``) -- the gradient-guided transfer;
        ``frozen``: names of parameters that do not train (train.py:189-220): their gradient launches are not emitted."""
        self.train = True
        super().__init__(m, w, B, H, W, device)
        self._init_train(tw, params, grads, input_grad, frozen)

    def _init_train(self, tw, params, grads, input_grad, frozen=()):
        m, w, B, H, W = self.m, self.w, self.B, self.H, self.W
        self.tw, self.params, self.grads = tw, params, grads
        self.param_grads, self.input_grad = grads is not None, input_grad
        self.frozen = frozenset(frozen)
        # the time-embedding chain (d proj -> time_emb_proj -> time_embedding MLP -> class table) runs iff one of its parameters trains
        self._temb_trains = self.param_grads and any(
            n not in self.frozen for n in (grads or {}) if ".time_emb_proj." in n or n.startswith("time_embedding.") or n.startswith("class_embedding."))
        # class_embed_type "timestep" (cond_unet_2d.py:146-153, 301-305): the class MLP's forward keeps its sinusoid / pre-activation rows
        # for the three gradient launches `_temb_bwd` adds (round 6).  "identity": the rows are an INPUT (nothing to differentiate)
        self._class_mlp = getattr(w, "class_mode", None) == "timestep"
        self._class_mlp_ran = False
        self.dsample = self._f32(B, m.config.in_channels, H, W) if input_grad else None
        if self.param_grads:
            self._check_layout()
        c0, tdim = m.config.block_out_channels[0], m.time_embed_dim
        self.t_feat, self.t_z1, self.t_emb = self._f32(B, c0), self._f32(B, tdim), self._f32(B, tdim)
        if self._class_mlp:
            self.c_feat, self.c_z1, self.c_emb, self.c_scratch = self._f32(B, c0), self._f32(B, tdim), self._f32(B, tdim), self._f32(B, w.proj_dim)
        self.temb_table = self._f32(B, w.proj_dim)
        self.bwd_ops: List[_Op] = []
        self.grad_ready: Dict[str, int] = {}   # parameter name -> index of the last backward op that writes its gradient
        self._emb_grad_op = None
        self._gact = {}
        self._fused_sums = {}      # id(gradient buffer) -> (per-split channel sums [B][splits][C], splits) left by its last writer
        self._sum_owner = {}       # id(activation) -> (GnBwdArgs, field) currently emitting those sums
        self._tmp_cache = {}
        self._wgrad_args = []
        self._sample_ptr_args = []
        self._build_backward()

    # ---- layout checks ---------------------------------------------------------------------------
    def _check_layout(self):
        for d in (self.params, self.grads):
            res = [n for n, mod in self.m.named_modules() if isinstance(mod, _Resnet)]
            for suffix in ("weight", "bias"):
                for a, b in zip(res[:-1], res[1:]):
                    if not _contiguous_after(d[f"{a}.time_emb_proj.{suffix}"], d[f"{b}.time_emb_proj.{suffix}"]):
                        raise ValueError("time_emb_proj parameters must be stacked contiguously (use training_param_order)")
            for n, mod in self.m.named_modules():
                if isinstance(mod, _Attention):
                    for suffix in ("weight", "bias"):
                        q, k, v = (d[f"{n}.{x}.{suffix}"] for x in ("to_q", "to_k", "to_v"))
                        if not (_contiguous_after(q, k) and _contiguous_after(k, v)):
                            raise ValueError("to_q/to_k/to_v parameters must be adjacent (use training_param_order)")
            for t in d.values():
                if t.dtype != torch.float32 or not t.is_contiguous() or t.device != torch.device(self.device):
                    raise ValueError("training parameters / gradients must be contiguous fp32 tensors on the plan's device")

    # ---- forward ---------------------------------------------------------------------------------
    def forward(self, sample: torch.Tensor, timesteps: torch.Tensor, labels: Optional[torch.Tensor],
                class_emb: Optional[torch.Tensor], out: torch.Tensor, stream):
        """One training forward: fp32 NCHW ``sample`` -> fp32 NCHW ``out``; keeps what ``backward`` needs."""
        a = self.temb_args
        a.rows = self.B
        if getattr(self.w, "class_mode", None) == "identity":      # the "labels" ARE the embedding rows (temb_rows does the same)
            if class_emb is None and labels is not None:
                class_emb = labels.to(dtype=torch.float32).contiguous()
            labels = None
            if class_emb is not None and class_emb.numel() != self.B * self.m.time_embed_dim:
                raise ValueError(f"class_embed_type='identity': {class_emb.numel()} elements for {self.B} rows of {self.m.time_embed_dim}")
        self._class_mlp_ran = False
        if self._class_mlp:
            # class_embedding(time_proj(class_labels)) (cond_unet_2d.py:301-305) through pd_temb with the class MLP's weights, keeping
            # the sinusoid rows and the pre-activation for its backward; rows given directly (class_emb: the unconditional steps'
            # zeros, utils_training.py:507-515) bypass the MLP, which then receives no gradient
            if class_emb is None and labels is not None:
                w, c = self.w, self.m.config
                vals = labels.to(dtype=torch.float32).contiguous()
                ca = L.TembArgs(rows=self.B, c0=c.block_out_channels[0], tdim=self.m.time_embed_dim, proj_dim=w.proj_dim,
                                flip_sin_to_cos=int(c.flip_sin_to_cos), freq_shift=float(c.freq_shift), num_classes=0,
                                timesteps=vals.data_ptr(), labels=None, class_emb=None, w1=w.cw1T.data_ptr(), b1=w.cb1.data_ptr(),
                                w2=w.cw2T.data_ptr(), b2=w.cb2.data_ptr(), class_table=None, wp=w.wpT.data_ptr(), bp=w.bp.data_ptr(),
                                emb=self.c_emb.data_ptr(), proj=self.c_scratch.data_ptr(), feat=self.c_feat.data_ptr(), z1=self.c_z1.data_ptr())
                L.check(self.lib.pd_temb(C.byref(ca), stream), "pd_temb (class MLP)")
                class_emb, self._class_mlp_ran, self._class_vals = self.c_emb, True, vals
            labels = None
        a.timesteps, a.labels, a.class_emb = timesteps.data_ptr(), L.ptr(labels), L.ptr(class_emb)
        a.emb, a.proj = self.t_emb.data_ptr(), self.temb_table.data_ptr()
        a.feat, a.z1 = self.t_feat.data_ptr(), self.t_z1.data_ptr()
        L.check(self.lib.pd_temb(C.byref(a), stream), "pd_temb")
        a.feat, a.z1 = None, None
        self.run(sample.data_ptr(), self.temb_table.data_ptr(), out.data_ptr(), stream)
        self._labels = labels
        src = getattr(self, "_centered", None)        # center_input_sample: what conv_in multiplied is 2 x - 1
        for args in self._sample_ptr_args:
            args.x = src.data_ptr() if src is not None else sample.data_ptr()
        self.keepalive = (sample, timesteps, labels, class_emb, out)

    # ---- backward emitters -----------------------------------------------------------------------
    def _g(self, act):
        """[gradient buffer of an activation, already holds a contribution?]"""
        e = self._gact.get(id(act))
        if e is None:
            t = torch.empty_like(act)
            self.bufs.append(t)
            e = [t, False]
            self._gact[id(act)] = e
        return e

    def _tmp(self, shape, tag, dtype=None):
        key = (tuple(shape), tag, dtype or self.tdt)
        t = self._tmp_cache.get(key)
        if t is None:
            t = torch.empty(shape, dtype=dtype or self.tdt, device=self.device)
            self.bufs.append(t)
            self._tmp_cache[key] = t
        return t

    def _drop_fused_sums(self, gbuf):
        """Another kind of launch writes this gradient buffer after a GroupNorm backward did: its channel sums are stale."""
        if self._fused_sums.pop(id(gbuf), None) is not None:
            for key, (args, fld) in list(self._sum_owner.items()):
                if self._gact.get(key, (None,))[0] is gbuf:
                    setattr(args, fld, None)
                    del self._sum_owner[key]

    def _b(self, fn, args, what, flops=0.0, nbytes=0.0):
        self.bwd_ops.append(_Op(fn, args, what, flops, nbytes))

    def _G(self, name, span=()):
        """Gradient tensor of a parameter, noting that the NEXT emitted op writes it (and the parameters fused behind it:
        ``span``) -- the schedule the overlapped data-parallel all-reduce follows.  None when the parameter (and everything fused
        behind it) is frozen: the emitters then skip the launch (a fused launch with one trainable member still runs and writes the
        frozen members' segments too -- ``FlatAdamWEMA.step`` zeroes those before the norm)."""
        if not self.param_grads:
            return None
        names = (name,) + tuple(span)
        if all(n in self.frozen for n in names):
            return None
        for n in names:
            if n not in self.frozen:
                self.grad_ready[n] = len(self.bwd_ops)
        return self.grads[name]

    def _G2(self, wname, bname):
        """(weight, bias) gradient tensors of a normalisation layer: both, or neither when both are frozen (one launch writes the two;
        a frozen member of a half-frozen pair is written too and zeroed by ``FlatAdamWEMA.step``)."""
        if not self.param_grads or (wname in self.frozen and bname in self.frozen):
            return None, None
        for n in (wname, bname):
            if n not in self.frozen:
                self.grad_ready[n] = len(self.bwd_ops)
        return self.grads[wname], self.grads[bname]

    def _esz(self):
        return 4 if self.code == L.PD_F32 else 2

    def _bias_grad(self, dy, total, valid=None, per_sample=None, per_stride=None):
        if not self.param_grads:
            return
        if total is None:                       # a frozen bias
            if per_sample is None or not self._temb_trains:
                return
            total = self._tmp((dy.shape[3],), "frozen_bias_sink", torch.float32)     # the per-sample sums still feed the time-embedding chain
        B, h, w, ch = dy.shape
        out = per_sample if per_sample is not None else self._tmp((B, ch), "chsum", torch.float32)
        fused = self._fused_sums.get(id(dy))
        if fused is not None:
            # the GroupNorm backward that stored the final value of this gradient also left its per-split channel sums
            ws, splits = fused[0], fused[1]
            # (round 6) sums in a workspace no other launch reuses, no per-sample consumer: the fold may run on the second stream with the slab folds
            side = _WGRAD_SIDE and per_sample is None and len(fused) > 2 and fused[2]
            if side:
                out = self._tmp((B, ch), "chsum_side", torch.float32)
            a = L.ChannelSumArgs(dtype=self.code, B=B, HW=h * w, C=ch, x=None, out=out.data_ptr(), out_stride=per_stride or ch,
                                 accumulate=0, total=total.data_ptr(), total_valid=valid or ch, workspace=ws.data_ptr(),
                                 splits=splits)
            self._b(self.lib.pd_channel_sum, a, "channel_sum_fused", 0.0, B * splits * ch * 4.0)
            self.bwd_ops[-1].side = bool(side)
            return
        splits = max(1, min(64, (h * w) // 64))
        ws = self._tmp((B * splits * ch,), "chsum_ws", torch.float32)
        a = L.ChannelSumArgs(dtype=self.code, B=B, HW=h * w, C=ch, x=dy.data_ptr(), out=out.data_ptr(),
                             out_stride=per_stride or ch, accumulate=0, total=total.data_ptr(), total_valid=valid or ch,
                             workspace=ws.data_ptr(), splits=splits)
        self._b(self.lib.pd_channel_sum, a, "channel_sum", 0.0, dy.numel() * self._esz())

    def _wgrad(self, x0, x1, gn, silu, dy, dw, *, ksize=3, stride=1, pad=1, upsample=0, cout_valid=0, cin_valid=0, phase=0):
        """Weight gradient of a convolution (``pd_conv_wgrad``); plain Linear layers (1x1, one dense source, no fused GroupNorm)
        go through the token-reduction GEMM ``pd_token_wgrad``."""
        if not self.param_grads or dw is None:          # (dw None: a frozen weight)
            return
        if (ksize == 1 and gn is not None and x1 is None and not cout_valid and not cin_valid and x0.shape[3] % 8 == 0 and dy.shape[3] % 8 == 0
                and not _NO_LINEAR_GRADS and not _NO_PREAPPLY_WGRAD):
            # a 1x1 layer behind a GroupNorm (the attention's fused q/k/v projection, Transformer2DModel.proj_in): pd_conv_wgrad's 1x1
            # form rebuilds the normalised input while staging and runs at ~190 TF/s; materialising it once (pd_gn_apply: one
            # bandwidth-bound pass over a tensor 1/16 .. 1/64 of the image-resolution ones) lets the token-reduction GEMM take it
            first = len(self.bwd_ops)
            ops, self.ops = self.ops, self.bwd_ops
            try:
                x0 = self._gn_apply(x0, None, gn, silu)
            finally:
                self.ops = ops
            self.bwd_ops[-1].what = "gn_apply_bwd"
            gn, silu = None, 0
            # `dw` came from _G(): "the NEXT emitted op writes this gradient" -- that op is now the pd_token_wgrad below, not the
            # pd_gn_apply just emitted (the overlapped all-reduce would hand the bucket over one launch early: with two real
            # ranks the reduced stale values then overwrite the gradient -- tests/test_gpu_two_rank_overlap.py)
            for n, r in self.grad_ready.items():
                if r >= first:
                    self.grad_ready[n] = len(self.bwd_ops)
        if (ksize == 1 and gn is None and x1 is None and not cout_valid and not cin_valid and x0.shape[3] % 8 == 0 and dy.shape[3] % 8 == 0
                and not _NO_LINEAR_GRADS):
            B, h, w, K = x0.shape
            N, M = dy.shape[3], B * h * w
            a = L.TokenWgradArgs(dtype=self.code, M=M, K=K, N=N, x=x0.data_ptr(), x_stride=K, dy=dy.data_ptr(), dy_stride=N,
                                 dw=dw.data_ptr(), accumulate=1)
            self._emit_wgrad(self.lib.pd_token_wgrad, a, "wgrad_linear", 2.0 * M * K * N, (M * (K + N)) * self._esz() + K * N * 4.0, self._twgrad_args)
            return
        B, hin, win, c0 = x0.shape
        c1 = x1.shape[3] if x1 is not None else 0
        _, hout, wout, cout = dy.shape
        if phase:          # sub-pixel phase of an upsampler (pd_wgrad_args.phase): dy is the gradient of the UPSAMPLED output, the grid is the low-resolution one
            hout, wout, ksize, pad = hin, win, 2, 0
        a = L.WgradArgs(dtype=self.code, B=B, Hin=hin, Win=win, Hout=hout, Wout=wout, C0=c0, C1=c1, Cout=cout, ksize=ksize,
                        stride=stride, pad=pad, upsample=upsample, silu=silu, x0=x0.data_ptr(), x1=L.ptr(x1),
                        scale=L.ptr(gn[0]) if gn else None, shift=L.ptr(gn[1]) if gn else None, dy=dy.data_ptr(),
                        dw=dw.data_ptr(), Cout_valid=cout_valid, Cin_valid=cin_valid, accumulate=1, phase=phase)
        flops = 2.0 * B * hout * wout * cout * (c0 + c1) * ksize * ksize
        nbytes = (x0.numel() + (x1.numel() if x1 is not None else 0) + dy.numel()) * self._esz() + dw.numel() * 4
        self._emit_wgrad(self.lib.pd_conv_wgrad, a, f"wgrad{ksize}x{ksize}", flops, nbytes, self._wgrad_args)

    def _emit_wgrad(self, fn, a, what, flops, nbytes, arglist):
        """One weight-gradient launch.  Side mode (``_WGRAD_SIDE``): two ops -- stage 1 (the GEMM) in the main sequence and stage 2 (the fold of its
        slab into the gradient) flagged ``side``, which :meth:`backward` runs on the plan's second stream; the gradient is final after the fold."""
        if not _WGRAD_SIDE:
            arglist.append((a, None))
            self._b(fn, a, what, flops, nbytes)
            return
        a.stage = 1
        a2 = type(a)()
        C.memmove(C.byref(a2), C.byref(a), C.sizeof(a))
        a2.stage = 2
        arglist.append((a, a2))
        self._b(fn, a, what, flops, nbytes)
        main = len(self.bwd_ops) - 1
        self._b(fn, a2, what + "_fold", 0.0, 0.0)
        self.bwd_ops[-1].side = True
        for n, r in self.grad_ready.items():        # "the op at `main` writes this gradient" -> its fold does
            if r == main:
                self.grad_ready[n] = main + 1

    def _dgrad(self, dy, wpk, cout, *, ksize=3, zero_stuff=False, into=None, tag="dz"):
        """Input gradient of a convolution: ``pd_conv`` over dy with the transposed/flipped weights (1x1: the GEMM kernel
        ``pd_linear`` on the same packed weights).  ``into`` = [buffer, initialised]: write (or accumulate, through ``residual``)
        straight into a gradient buffer."""
        B, h, w, cin = dy.shape
        ho, wo = (2 * h, 2 * w) if zero_stuff else (h, w)
        if into is not None:
            y, res = into[0], (into[0] if into[1] else None)
            into[1] = True
            self._drop_fused_sums(into[0])
        else:
            y, res = self._tmp((B, ho, wo, cout), tag), None
        ops, self.ops = self.ops, self.bwd_ops
        try:
            if ksize == 1 and not zero_stuff and cin % 32 == 0 and cout % 8 == 0 and not _NO_LINEAR_GRADS:
                self._linear(dy, wpk, self._zero_bias, cout, residual=res, y=y, what="dgrad_linear")
            else:
                # zero_stuff: True / 2 = samples at the even positions (stride-2 pad-1 forward), 3 = at the odd ones (pad-0 forward)
                self._conv(dy, None, wpk, self._zero_bias, cout, ksize=ksize, pad=ksize // 2,
                           upsample=(2 if zero_stuff is True else int(zero_stuff)) if zero_stuff else 0, residual=res, y=y, stats=False)
                self.bwd_ops[-1].what = f"dgrad{ksize}x{ksize}"
        finally:
            self.ops = ops
        return y

    def _gn_bwd(self, gn, dz, silu, *, combined=False, res=None, wname=None, mod_off=None):
        """``mod_off``: column of this norm's [scale | shift] block in the step's projection table (scale_shift ResNet blocks,
        pd_gn_bwd_args.mod): the backward then also writes d [scale | shift] into the same columns of ``self.dproj``."""
        s = self.gn_saved[id(gn[0])]
        x0, x1 = s.x0, s.x1
        B, h, w, c0 = x0.shape
        c1 = x1.shape[3] if x1 is not None else 0
        g0 = self._g(x0)
        g1 = self._g(x1) if x1 is not None else None
        splits = max(1, min(64, (h * w) // 64, -(-1024 // B)))
        partial = self._tmp((B * splits * (c0 + c1) * 2,), "gnpart", torch.float64)
        coef = self._tmp((B, self.groups, 2), "gncoef", torch.float32)
        dgam, dbet = self._G2(wname + ".weight", wname + ".bias")
        a = L.GnBwdArgs(dtype=self.code, B=B, HW=h * w, C0=c0, C1=c1, groups=self.groups, silu=silu, x0=x0.data_ptr(),
                        x1=L.ptr(x1), dz0=dz.data_ptr(), dz1=None, mean=s.mean.data_ptr(), rstd=s.rstd.data_ptr(),
                        gamma=s.gamma.data_ptr(), beta=s.beta.data_ptr(), partial=partial.data_ptr(), splits=splits,
                        coef=coef.data_ptr(), dx0=g0[0].data_ptr(), dx1=(g1[0].data_ptr() if g1 else None),
                        accumulate0=int(g0[1]), accumulate1=int(g1[1]) if g1 else 0,
                        dgamma=L.ptr(dgam), dbeta=L.ptr(dbet),
                        dz_combined=1 if (combined and c1) else 0, res=L.ptr(res))
        if mod_off is not None:
            a.mod, a.mod_stride = self.temb_table.data_ptr() + 4 * mod_off, self.w.proj_dim
            a.dmod = self.dproj.data_ptr() + 4 * mod_off if self._temb_trains else None
        g0[1] = True
        if g1:
            g1[1] = True
        # this launch is (so far) the last writer of the gradients of x0 / x1: it also emits their per-split channel sums,
        # which the producer block's bias / time-embedding gradients read instead of another pass (a later writer of the
        # same buffer re-registers and supersedes these)
        for src, fld, gb in ((x0, "sum0", g0), (x1, "sum1", g1)):
            if src is None or not self.param_grads:
                continue
            prev = self._sum_owner.pop(id(src), None)
            if prev is not None:
                setattr(prev[0], prev[1], None)            # superseded: that launch no longer needs to emit sums
            st = torch.empty((B, splits, src.shape[3]), dtype=torch.float32, device=self.device)
            self.bufs.append(st)
            setattr(a, fld, st.data_ptr())
            self._sum_owner[id(src)] = (a, fld)
            self._fused_sums[id(gb[0])] = (st, splits, True)       # (`st` belongs to this launch alone)
        n = B * h * w * (c0 + c1)
        self._b(self.lib.pd_gn_silu_bwd, a, "gn_silu_bwd", 0.0, n * self._esz() * (5 + (1 if res is not None else 0)))

    # ---- backward plan ---------------------------------------------------------------------------
    def _zero_bias_len(self):
        return max(max(self.m.config.block_out_channels) * 3, 64) + 64

    def _build_backward(self):
        w = self.w
        self._zero_bias = torch.zeros(self._zero_bias_len(), dtype=torch.float32, device=self.device)
        self.dproj = self._f32(self.B, w.proj_dim)
        self._twgrad_args = []
        for rec in reversed(self.tape):
            self._bwd_record(rec)
        if not self.param_grads:
            return
        self._temb_bwd()
        # slabs of the weight-gradient launches: ONE per launch in side mode (its fold runs later, on the second stream), else one shared by all
        # (they then run back to back on one stream)
        for pairs, ws in ((self._wgrad_args, self.lib.pd_conv_wgrad_workspace), (self._twgrad_args, self.lib.pd_token_wgrad_workspace)):
            if not pairs:               # (no convolution weight gradients when every convolution is frozen: attention-only fine-tuning)
                continue
            if _WGRAD_SIDE:
                for a, a2 in pairs:
                    need = ws(C.byref(a))
                    slab = torch.empty(need // 4 + 16, dtype=torch.float32, device=self.device)
                    self.bufs.append(slab)
                    a.slab = a2.slab = slab.data_ptr()
                    a.slab_bytes = a2.slab_bytes = need
            else:
                need = max(ws(C.byref(a)) for a, _ in pairs)
                slab = torch.empty(need // 4 + 16, dtype=torch.float32, device=self.device)
                self.bufs.append(slab)
                for a, _ in pairs:
                    a.slab, a.slab_bytes = slab.data_ptr(), need

    def _bwd_record(self, rec):
        """Emit the backward launches of one forward tape record."""
        m, w, tw, c = self.m, self.w, self.tw, self.m.config
        G = self._G
        B, H, W = self.B, self.H, self.W
        boc0 = c.block_out_channels[0]
        k = rec.kind
        if k == "conv_out":
            dy = self._tmp((B, H, W, w.conv_out_pad), "dy_out")
            a = L.NchwToNhwcArgs(dtype=self.code, B=B, C=c.out_channels, HW=H * W, Cpad=w.conv_out_pad, x=None,
                                 out=dy.data_ptr())
            self._dout_args = a
            self._b(self.lib.pd_nchw_to_nhwc, a, "nchw_to_nhwc", 0.0, B * H * W * c.out_channels * 4.0)
            self._bias_grad(dy, G("conv_out.bias"), valid=c.out_channels)
            self._wgrad(rec.x, None, rec.gn, 1, dy, G("conv_out.weight"), cout_valid=c.out_channels)
            dz = self._dgrad(dy, tw.conv_out_d, boc0)
            self._gn_bwd(rec.gn, dz, 1, wname="conv_norm_out")
        elif k == "resnet":
            self._resnet_bwd(rec)
        elif k == "attn":
            self._attn_bwd(rec)
        elif k == "attn_nhwc":
            self._attn_nhwc_bwd(rec)
        elif k == "down":
            # Downsample2D(padding=0) pads (0, 1, 0, 1) and convolves without padding (orig_google_ddpm_model_denoiser.json):
            # y[o] = sum_k w[k] x[2 o + k], so its input gradient is the pad-1 convolution over dY zero-stuffed at the ODD positions
            pad = rec.e.padding
            if pad not in (0, 1):
                raise NotImplementedError(f"training: Downsample2D with padding {pad}")
            dout = self._g(rec.out)[0]
            self._bias_grad(dout, G(rec.name + ".conv.bias"))
            self._wgrad(rec.x, None, None, 0, dout, G(rec.name + ".conv.weight"), stride=2, pad=pad)
            self._dgrad(dout, tw.samplers[rec.name].wd, rec.x.shape[3], zero_stuff=(2 if pad == 1 else 3), into=self._g(rec.x))
        elif k == "up":
            dout = self._g(rec.out)[0]
            self._bias_grad(dout, G(rec.name + ".conv.bias"))
            gx = self._g(rec.x)
            _, h, ww, ch = rec.x.shape
            wd4 = getattr(tw.samplers[rec.name], "wd4", None)
            sub = wd4 is not None and self._subpixel_up_ok(rec.x)
            if sub:        # the weight gradient through the four phases as well: each adds its 2x2 tap gradients to the 3x3 taps they sum (4 / 9 of the FLOPs)
                gw = G(rec.name + ".conv.weight")
                for ph in range(4):
                    self._wgrad(rec.x, None, None, 0, dout, gw, phase=1 + ph)
                if gw is not None:      # the gradient is final after the LAST of the four launches (G() recorded the first: the overlapped all-reduce hands its bucket over there)
                    self.grad_ready[rec.name + ".conv.weight"] = len(self.bwd_ops) - 1
            else:
                self._wgrad(rec.x, None, None, 0, dout, G(rec.name + ".conv.weight"), upsample=1)
            if sub:
                # d x = sum over the four phases of a 2x2 convolution over that phase's pixels of d out (transposed, flipped phase kernels):
                # each launch accumulates into the gradient of the low-resolution tensor through `residual`
                for ph in range(4):
                    a = L.ConvArgs(dtype=self.code, B=B, Hin=h, Win=ww, Hout=h, Wout=ww, C0=ch, C1=0, Cout=ch, Cout_pad=ch, ksize=2, stride=1, pad=0,
                                   upsample=0, silu=0, out_mode=L.PD_OUT_NHWC, heads=0, x0=dout.data_ptr(), x1=None, scale=None, shift=None,
                                   w_packed=wd4[ph].data_ptr(), bias=self._zero_bias.data_ptr(), temb=None, temb_stride=self.w.proj_dim,
                                   residual=(gx[0].data_ptr() if (gx[1] or ph > 0) else None), y=gx[0].data_ptr(), stats_out=None, im2col3=0,
                                   tail_x0=None, tail_x1=None, tail_C0=0, tail_C1=0, phase=1 + ph, phase_in=1)
                    self._b(self.lib.pd_conv, a, "dgrad3x3", 2.0 * B * 4 * h * ww * ch * ch * 9 / 4.0,
                            (B * 4 * h * ww * ch / 4.0 + 2.0 * B * h * ww * ch) * self._esz() + ch * ch * 4 * self._esz())
                gx[1] = True
                self._drop_fused_sums(gx[0])
            else:
                du = self._dgrad(dout, tw.samplers[rec.name].wd, rec.x.shape[3], tag="du")
                a = L.Pool2x2Args(dtype=self.code, B=B, H=h, W=ww, C=ch, du=du.data_ptr(), dx=gx[0].data_ptr(),
                                  accumulate=int(gx[1]))
                gx[1] = True
                self._drop_fused_sums(gx[0])
                self._b(self.lib.pd_pool2x2_sum, a, "pool2x2", 0.0, du.numel() * self._esz() * 1.25)
        elif k == "conv_in":
            dout = self._g(rec.out)[0]
            if self.input_grad:
                ops, self.ops = self.ops, self.bwd_ops
                try:
                    self._conv(dout, None, tw.conv_in_d, self._zero_bias, c.in_channels, out_mode=L.PD_OUT_NCHW_F32,
                               cout_pad=32, y=self.dsample, stats=False)
                finally:
                    self.ops = ops
                self.bwd_ops[-1].what = "dgrad3x3"
                if getattr(self, "_centered", None) is not None:      # d (2 x - 1) / d x = 2 (cond_unet_2d.py:272-273)
                    ones, two, _ = self._center_const
                    zero = torch.zeros_like(two)
                    self.bufs.append(zero)
                    sa = L.AddNoiseArgs(numel=self.dsample.numel(), per_sample=self.dsample[0].numel(), velocity=0, x=self.dsample.data_ptr(),
                                        noise=ones.data_ptr(), sa=two.data_ptr(), sb=zero.data_ptr(), out=self.dsample.data_ptr())
                    self._b(self.lib.pd_add_noise, sa, "center_bwd", 0.0, 2.0 * self.dsample.numel() * 4)
            if not self.param_grads:
                return
            self._bias_grad(dout, G("conv_in.bias"))
            if "conv_in.weight" in self.frozen:
                return
            cols = self._tmp((B, H, W, 32), "im2col")
            a = L.Im2col3Args(dtype=self.code, B=B, H=H, W=W, C=c.in_channels, x=None, out=cols.data_ptr())
            self._sample_ptr_args.append(a)
            self._b(self.lib.pd_im2col3, a, "im2col3", 0.0, cols.numel() * self._esz())
            self._wgrad(cols, None, None, 0, dout, G("conv_in.weight"), ksize=1, pad=0, cin_valid=c.in_channels * 9)
        else:
            raise NotImplementedError(f"backward of tape record {k!r}")

    def _resnet_bwd(self, rec):
        e, te = rec.e, self.tw.resnets[rec.name]
        G = self._G
        n = rec.name
        x0, x1 = rec.x0, rec.x1
        cin = e.cin
        dout = self._g(rec.out)[0]
        self._bias_grad(dout, G(n + ".conv2.bias"))
        if rec.z2 is not None:
            self._wgrad(rec.z2, None, None, 0, dout, G(n + ".conv2.weight"))
        else:
            self._wgrad(rec.h1, None, rec.gn2, 1, dout, G(n + ".conv2.weight"))
        if e.fused_shortcut:
            self._bias_grad(dout, G(n + ".conv_shortcut.bias"))
            self._wgrad(x0, x1, None, 0, dout, G(n + ".conv_shortcut.weight"), ksize=1, pad=0)
            res = self._dgrad(dout, te.wsd, cin, ksize=1, tag="dshort")
        else:
            res = dout
        dz2 = self._dgrad(dout, te.w2d, e.cout)
        ss = getattr(e, "scale_shift", False)
        # "scale_shift" (cond_unet_2d.py:180,191,225): the projection modulates norm2's affine, h = norm2(h1) (1 + scale) + shift; its
        # gradient [d scale | d shift] comes out of the GroupNorm backward (pd_gn_bwd_args.mod / dmod) instead of conv1's epilogue
        self._gn_bwd(rec.gn2, dz2, 1, wname=n + ".norm2", mod_off=e.temb_off if ss else None)
        dh1 = self._g(rec.h1)[0]
        if ss:
            self._bias_grad(dh1, G(n + ".conv1.bias"))
        else:
            # d time_emb_proj output [n][co] = sum over pixels of d h1 (the projection is broadcast over the pixels)
            per = self.dproj[:, e.temb_off:]
            self._bias_grad(dh1, G(n + ".conv1.bias"), per_sample=per, per_stride=self.w.proj_dim)
        if rec.z1 is not None:
            self._wgrad(rec.z1, None, None, 0, dh1, G(n + ".conv1.weight"))
        else:
            self._wgrad(x0, x1, rec.gn1, 1, dh1, G(n + ".conv1.weight"))
        dz1 = self._dgrad(dh1, te.w1d, cin, tag="dz1")
        self._gn_bwd(rec.gn1, dz1, 1, combined=True, res=res, wname=n + ".norm1")

    def _attn_bwd(self, rec):
        e, te, n = rec.e, self.tw.attns[rec.name], rec.name
        G = self._G
        B, h, w, ch = rec.x.shape
        N = h * w
        dout = self._g(rec.out)[0]
        self._bias_grad(dout, G(n + ".to_out.0.bias"))
        self._wgrad(rec.o, None, None, 0, dout, G(n + ".to_out.0.weight"), ksize=1, pad=0)
        do = self._dgrad(dout, te.wod, ch, ksize=1, tag="do")
        dqkv = self._tmp((B, h, w, 3 * ch), "dqkv")
        delta = self._tmp((B, e.heads, N), "delta", torch.float32)
        a = L.AttnBwdArgs(dtype=self.code, B=B, heads=e.heads, N=N, q=rec.qkv[0].data_ptr(), k=rec.qkv[1].data_ptr(),
                          v=rec.qkv[2].data_ptr(), o=rec.o.data_ptr(), dout=do.data_ptr(), lse=rec.lse.data_ptr(),
                          delta=delta.data_ptr(), dqkv=dqkv.data_ptr())
        # one-pass backward (round 5; 16-bit engines, N >= 512): the partial dQ of every 512-key block, fp32, in a workspace all attention
        # layers of the step share.  OPT-IN (PD_ATTN_BWD_FUSED=1): 4-5 % faster than the two kernels as an op (1.21 vs 1.28 ms per
        # configs[1] layer) and neutral-to-slower on the whole step (2 324 vs 2 339 images/s, 4 alternating rounds:
        # profiles/r5_ab_attn_bwd_one_pass.log) -- 200 registers hold it at 2 waves per SIMD, where matrix and vector work do not overlap
        need = int(self.lib.pd_attn_d8_bwd_workspace(C.byref(a))) if __import__("os").environ.get("PD_ATTN_BWD_FUSED", "0") == "1" else 0
        if need > 0:
            slab = self._tmp((need // 4,), "attn_dq_slab", torch.float32)
            a.slab, a.slab_bytes = slab.data_ptr(), need
        self._b(self.lib.pd_attn_d8_bwd, a, "attn_d8_bwd", 10.0 * B * e.heads * N * N * 8, 8.0 * B * N * ch * self._esz())
        self._bias_grad(dqkv, G(n + ".to_q.bias", (n + ".to_k.bias", n + ".to_v.bias")))                       # [dq | dk | dv] biases are adjacent
        self._wgrad(rec.x, None, rec.gn, 0, dqkv, G(n + ".to_q.weight", (n + ".to_k.weight", n + ".to_v.weight")), ksize=1, pad=0)
        dz = self._dgrad(dqkv, te.wqkvd, ch, ksize=1, tag="dzattn")
        self._gn_bwd(rec.gn, dz, 0, res=dout, wname=n + ".group_norm")

    def _attn_nhwc_bwd(self, rec):
        """Backward of ``UNetPlan._attn_nhwc`` (head_dim 64, or one wide head of 128 / 256 / 512 channels: ``attention_head_dim``
        null of orig_google_ddpm_model_denoiser.json): q | k | v live NHWC in one [B][N][3C] tensor, so the attention gradient
        is written straight into the fused projection's output gradient [dq | dk | dv]."""
        e, te, n = rec.e, self.tw.attns[rec.name], rec.name
        G = self._G
        B, h, w, ch = rec.x.shape
        N, esz, d = h * w, self._esz(), rec.d
        dout = self._g(rec.out)[0]
        self._bias_grad(dout, G(n + ".to_out.0.bias"))
        self._wgrad(rec.o, None, None, 0, dout, G(n + ".to_out.0.weight"), ksize=1, pad=0)
        do = self._dgrad(dout, te.wod, ch, ksize=1, tag="do")
        dqkv = self._tmp((B, h, w, 3 * ch), "dqkv")
        delta = self._tmp((B, e.heads, N), "delta", torch.float32)
        p, dp = rec.qkv.data_ptr(), dqkv.data_ptr()
        common = dict(dtype=self.code, B=B, heads=e.heads, Nq=N, Nkv=N, q=p, q_stride=3 * ch, k=p + ch * esz, v=p + 2 * ch * esz,
                      kv_stride=3 * ch, o=rec.o.data_ptr(), dout=do.data_ptr(), o_stride=ch, lse=rec.lse.data_ptr(),
                      delta=delta.data_ptr(), dq=dp, dq_stride=3 * ch, dk=dp + ch * esz, dv=dp + 2 * ch * esz, dkv_stride=3 * ch)
        if d == 64:
            self._b(self.lib.pd_attn_d64_bwd, L.AttnD64BwdArgs(**common), "attn_d64_bwd", 10.0 * B * N * N * ch, 8.0 * B * N * ch * esz)
        else:
            self._b(self.lib.pd_attn_wide_bwd, L.AttnWideBwdArgs(D=d, scale=float(d) ** -0.5, **common), "attn_wide_bwd",
                    10.0 * B * N * N * ch, 8.0 * B * N * ch * esz)
        self._bias_grad(dqkv, G(n + ".to_q.bias", (n + ".to_k.bias", n + ".to_v.bias")))
        self._wgrad(rec.x, None, rec.gn, 0, dqkv, G(n + ".to_q.weight", (n + ".to_k.weight", n + ".to_v.weight")), ksize=1, pad=0)
        dz = self._dgrad(dqkv, te.wqkvd, ch, ksize=1, tag="dzattn")
        self._gn_bwd(rec.gn, dz, 0, res=dout, wname=n + ".group_norm")

    def _temb_bwd(self):
        if not self._temb_trains:
            return
        m, w, P = self.m, self.w, self.params
        # (this chain's launches write weight and bias of a layer together and the projections of all ResNet blocks as one stacked
        # matrix: a partially frozen chain runs whole, FlatAdamWEMA.step zeroes the frozen members' gradient segments)
        def G(name, span=()):
            for n in (name,) + tuple(span):
                if n not in self.frozen:
                    self.grad_ready[n] = len(self.bwd_ops)
            return self.grads[name]
        B, tdim, c0, pd = self.B, m.time_embed_dim, m.config.block_out_channels[0], w.proj_dim
        res = [n for n, mod in m.named_modules() if isinstance(mod, _Resnet)]
        first = res[0]
        lib = self.lib
        demb, dz1 = self._f32(B, tdim), self._f32(B, tdim)
        self._b(lib.pd_linear_wgrad, L.LinearWgradArgs(rows=B, in_dim=tdim, out_dim=pd, x_silu=1, dy=self.dproj.data_ptr(),
                x=self.t_emb.data_ptr(),
                dw=G(first + ".time_emb_proj.weight", [r + ".time_emb_proj.weight" for r in res[1:]]).data_ptr(),
                db=G(first + ".time_emb_proj.bias", [r + ".time_emb_proj.bias" for r in res[1:]]).data_ptr()), "linear_wgrad")
        self._b(lib.pd_linear_dgrad, L.LinearDgradArgs(rows=B, in_dim=tdim, out_dim=pd, dy=self.dproj.data_ptr(),
                w=P[first + ".time_emb_proj.weight"].data_ptr(), pre=self.t_emb.data_ptr(), dx=demb.data_ptr()), "linear_dgrad")
        if getattr(getattr(m, "class_embedding", None), "weight", None) is not None and "class_embedding.weight" not in self.frozen:
            self._emb_grad_args = L.EmbeddingGradArgs(rows=B, dim=tdim, num_classes=m.class_embedding.weight.shape[0], labels=None,
                                                      d=demb.data_ptr(), dtable=G("class_embedding.weight").data_ptr())
            self._emb_grad_at = len(self.bwd_ops)
        if self._class_mlp and any(n.startswith("class_embedding.") and n not in self.frozen for n in self.grads):
            # emb = time_embedding(...) + class_embedding(time_proj(labels)): the class MLP sees the same d emb.  These three launches run
            # only on steps whose labels went through the MLP (`backward` skips the [start, end) range otherwise)
            dcz1 = self._f32(B, tdim)
            start = len(self.bwd_ops)
            self._b(lib.pd_linear_wgrad, L.LinearWgradArgs(rows=B, in_dim=tdim, out_dim=tdim, x_silu=1, dy=demb.data_ptr(),
                    x=self.c_z1.data_ptr(), dw=G("class_embedding.linear_2.weight").data_ptr(),
                    db=G("class_embedding.linear_2.bias").data_ptr()), "linear_wgrad")
            self._b(lib.pd_linear_dgrad, L.LinearDgradArgs(rows=B, in_dim=tdim, out_dim=tdim, dy=demb.data_ptr(),
                    w=P["class_embedding.linear_2.weight"].data_ptr(), pre=self.c_z1.data_ptr(), dx=dcz1.data_ptr()), "linear_dgrad")
            self._b(lib.pd_linear_wgrad, L.LinearWgradArgs(rows=B, in_dim=c0, out_dim=tdim, x_silu=0, dy=dcz1.data_ptr(),
                    x=self.c_feat.data_ptr(), dw=G("class_embedding.linear_1.weight").data_ptr(),
                    db=G("class_embedding.linear_1.bias").data_ptr()), "linear_wgrad")
            self._class_mlp_ops = (start, len(self.bwd_ops))
        self._b(lib.pd_linear_wgrad, L.LinearWgradArgs(rows=B, in_dim=tdim, out_dim=tdim, x_silu=1, dy=demb.data_ptr(),
                x=self.t_z1.data_ptr(), dw=G("time_embedding.linear_2.weight").data_ptr(),
                db=G("time_embedding.linear_2.bias").data_ptr()), "linear_wgrad")
        self._b(lib.pd_linear_dgrad, L.LinearDgradArgs(rows=B, in_dim=tdim, out_dim=tdim, dy=demb.data_ptr(),
                w=P["time_embedding.linear_2.weight"].data_ptr(), pre=self.t_z1.data_ptr(), dx=dz1.data_ptr()), "linear_dgrad")
        self._b(lib.pd_linear_wgrad, L.LinearWgradArgs(rows=B, in_dim=c0, out_dim=tdim, x_silu=0, dy=dz1.data_ptr(),
                x=self.t_feat.data_ptr(), dw=G("time_embedding.linear_1.weight").data_ptr(),
                db=G("time_embedding.linear_1.bias").data_ptr()), "linear_wgrad")

    # ---- execution -------------------------------------------------------------------------------
    def backward(self, dout: torch.Tensor, stream, after_op=None):
        """Accumulate d loss / d parameters into ``grads`` given ``dout`` = d loss / d (UNet output), fp32 NCHW.
        ``after_op``: {op index: callable} run right after that op was enqueued (gradient buckets becoming final)."""
        self._dout_args.x = dout.data_ptr()
        byref, check = C.byref, L.check
        labels = getattr(self, "_labels", None)
        emb_at = getattr(self, "_emb_grad_at", -1)
        skip0, skip1 = getattr(self, "_class_mlp_ops", (0, 0)) if not getattr(self, "_class_mlp_ran", False) else (0, 0)
        side, ms, evs, join = self._side_streams(stream)
        if after_op is not None or _data_parallel():
            # data-parallel run (gradient buckets are handed to the exchange on its own stream): the folds stay on the main stream, right behind
            # their GEMMs -- the launch order of rounds 1-5, in every step of the run (so that a step without the exchange, bench.py's
            # `step_ms_no_comm`, is the same program).  (Two ranks on ONE card over gloo, the only N > 1 run this box allows, were 15 x slower with
            # three streams per process; nothing says RCCL on separate cards would be, nothing here can show it.)
            side = None
        k, dirty = 0, False
        for i, op in enumerate(self.bwd_ops):
            if skip0 <= i < skip1:            # class MLP (class_embed_type = "timestep") on a step whose rows bypassed it
                continue
            if i == emb_at and labels is not None:
                self._emb_grad_args.labels = labels.data_ptr()
                check(self.lib.pd_embedding_grad(byref(self._emb_grad_args), stream), "pd_embedding_grad")
            if op.side and side is not None:
                # the fold of the slab the previous op (its GEMM, on the main stream) just filled: second stream, behind an event
                ev = evs[k]
                k += 1
                ev.record(ms)
                side.wait_event(ev)
                rc = op.fn(byref(op.args), side.cuda_stream)
                dirty = True
            else:
                rc = op.fn(byref(op.args), stream)
            if rc:
                check(rc, op.what)
            if after_op is not None:
                f = after_op.get(i)
                if f is not None:
                    if dirty:                 # a gradient bucket becomes final here: the folds so far belong to it
                        join.record(side)
                        ms.wait_event(join)
                        dirty = False
                    f()
        if dirty:                             # whoever reads the gradients next does so on the main stream
            join.record(side)
            ms.wait_event(join)
        self._keep_dout = dout

    def _side_streams(self, stream):
        """(second stream, the main stream as a torch object, one event per side op, a join event) -- or (None, ...) when the plan has no side op."""
        st = getattr(self, "_side_state", None)
        if st is None:
            n = sum(1 for op in self.bwd_ops if op.side)
            if n == 0:
                st = self._side_state = (None, None, None, None, -1)
            else:
                with torch.cuda.device(self.device):
                    st = self._side_state = (torch.cuda.Stream(self.device), None, [torch.cuda.Event() for _ in range(n)], torch.cuda.Event(), -1)
        if st[0] is None:
            return None, None, None, None
        if st[4] != stream:                   # the main stream as a torch object (cached per handle)
            cur = torch.cuda.current_stream(self.device)
            ms = cur if cur.cuda_stream == stream else torch.cuda.ExternalStream(stream, device=self.device)
            st = self._side_state = (st[0], ms, st[2], st[3], stream)
        return st[0], st[1], st[2], st[3]


def _data_parallel() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def plan_grad_buckets(sizes: List[int], ready: List[int], bucket_elems: int) -> List[Tuple[int, int, int]]:
    """Cut a flat gradient buffer (parameters of ``sizes`` elements, in buffer order; gradient ``i`` final after backward
    op ``ready[i]``) into contiguous buckets of about ``bucket_elems`` elements.  Returns (start, end, ready_op) per bucket
    sorted by ``ready_op``: the order in which the buckets can be all-reduced while the backward is still running.
    Buckets are few and large on purpose: an xGMI ring all-reduce is per-link bound, so small messages waste it."""
    out, start, acc, rdy = [], 0, 0, -1
    off = 0
    for sz, r in zip(sizes, ready):
        off += sz
        acc += sz
        rdy = max(rdy, r)
        if acc >= bucket_elems:
            out.append((start, off, rdy))
            start, acc, rdy = off, 0, -1
    if acc:
        out.append((start, off, rdy))
    return sorted(out, key=lambda b: b[2])


def fuse_pack_jobs(jobs):
    """A weight is re-packed twice after every optimizer step -- the forward layout and the input-gradient layout (transposed, taps
    flipped) -- from the same fp32 master tensor.  Pairs whose 32 x 32 blocks coincide become ONE job with ``dst2`` (ABI 7): the master
    weights are read once (the re-pack of the SD-2.1 UNet: 6.9 -> 3.5 GB of reads per step).  Jobs without a partner stay as they are."""
    if os.environ.get("PD_PACK_FUSE", "1") == "0":      # diagnostic override (same-box A/B)
        return list(jobs)
    fwd = {}
    for a in jobs:
        if not a.dgrad and not a.dst2:
            fwd.setdefault((a.src, a.ksize, a.cout, a.cin, a.cout_pad, a.cin_pad, a.src_in), []).append(a)
    out, used = [], set()
    for a in jobs:
        if a.dgrad:
            cands = fwd.get((a.src, a.ksize, a.cin, a.cout, a.cin_pad, a.cout_pad, a.src_in), [])
            partner = next((f for f in cands if id(f) not in used), None)
            if partner is not None:
                used.add(id(partner))
                partner.dst2, partner.dst2_ct_stride = a.dst, a.dst_ct_stride
                continue
        out.append(a)
    return out


def run_pack_jobs(lib, jobs, stream, cache, device):
    """All ``pd_pack_weight`` jobs of an optimizer step as ONE ``pd_pack_weight_batch`` launch: the descriptors are uploaded
    once (``cache``: a dict owned by the re-packer) next to the block-range table the kernel searches."""
    if not jobs:
        return
    st = cache.get("batch")
    if st is None:
        jobs = fuse_pack_jobs(jobs)
        for a in jobs:     # what pd_pack_weight would refuse
            if a.cout_pad % 32 or a.cin_pad % 32 or a.cout_pad < a.cout or a.cin_pad < a.cin or a.dtype != jobs[0].dtype:
                raise L.PhenDiffHipError("pd_pack_weight_batch: inconsistent job descriptors")
        dev = torch.device(device)
        st = []
        # one launch per kernel size: the workgroup's LDS tile is sized by the launch's largest kernel (37 KB for 3x3 against 4 KB for the
        # Linear / 1x1 blocks, which are most of the latent-diffusion UNet's blocks and would run at a quarter of the occupancy beside them)
        for ks in sorted({a.ksize for a in jobs}):
            sel = [a for a in jobs if a.ksize == ks]
            raw = (L.PackWeightArgs * len(sel))(*sel)
            table = torch.frombuffer(bytearray(bytes(raw)), dtype=torch.uint8).to(dev)
            starts, tot = [0], 0
            for a in sel:
                tot += (a.cout_pad // 32) * (a.cin_pad // 32)
                starts.append(tot)
            starts_t = torch.tensor(starts, dtype=torch.int32, device=dev)
            args = L.PackWeightBatchArgs(dtype=sel[0].dtype, n=len(sel), jobs=table.data_ptr(), starts=starts_t.data_ptr(), total_blocks=tot,
                                         max_ksize=ks)
            st.append((args, table, starts_t))
        cache["batch"] = st
    for args, _, _ in st:
        L.check(lib.pd_pack_weight_batch(C.byref(args), stream), "pd_pack_weight_batch")


class _Repacker:
    """After an optimizer step: fp32 master parameters -> every kernel-layout copy the plans read (``_PackedWeights`` and
    ``TrainWeights`` tensors, IN PLACE), as ``pd_pack_weight`` launches plus a handful of small fp32 copies.  Parameters
    that the kernels read as plain fp32 vectors (GroupNorm affine, most biases, the class table) alias the flat master
    buffer and need nothing."""

    def __init__(self, m: CustomCondUNet2DModel, w: _PackedWeights, tw: TrainWeights):
        self.lib = L.lib()
        self.jobs, self.small, self.pre = [], [], []      # pre: torch-side preparations the pack jobs read (run first)
        self.jobs_device = m.conv_in.weight.device
        code = w.code

        def job(dst, src, cout, cin, k, *, dgrad=0, cout_pad=None, cin_pad=None, src_in=None, ct_stride=None, dst_off=0):
            cp = cout_pad or ((cout + 31) // 32) * 32
            ip = cin_pad or ((cin + 31) // 32) * 32
            per_ct = (ip // 32) * k * k * 2 * 64 * 8
            esz = dst.element_size()
            self.jobs.append(L.PackWeightArgs(dtype=code, cout=cout, cin=cin, cout_pad=cp, cin_pad=ip, ksize=k,
                                              src_in=src_in or (cout if dgrad else cin), dgrad=dgrad, src=src.data_ptr(),
                                              dst=dst.data_ptr() + dst_off * esz, dst_ct_stride=ct_stride or per_ct))

        ci = m.conv_in.weight.shape[1]
        job(w.conv_in_wv, m.conv_in.weight, m.conv_in.weight.shape[0], ci * 9, 1, cin_pad=32)
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
            elif isinstance(mod, _Attention):
                e, t = w.attns[name], tw.attns[name]
                ch = mod.to_q.weight.shape[0]
                if not (_contiguous_after(mod.to_q.weight.data, mod.to_k.weight.data)
                        and _contiguous_after(mod.to_k.weight.data, mod.to_v.weight.data)
                        and _contiguous_after(mod.to_q.bias.data, mod.to_k.bias.data)
                        and _contiguous_after(mod.to_k.bias.data, mod.to_v.bias.data)):
                    raise ValueError("to_q/to_k/to_v parameters must be adjacent (use training_param_order)")
                job(e.wqkv, mod.to_q.weight, 3 * ch, ch, 1)
                job(e.wo, mod.to_out[0].weight, ch, ch, 1)
                job(t.wqkvd, mod.to_q.weight, ch, 3 * ch, 1, dgrad=1)
                job(t.wod, mod.to_out[0].weight, ch, ch, 1, dgrad=1)
                qb, dst = mod.to_q.bias, e.bqkv
                self.small.append(lambda qb=qb, dst=dst, ch=ch: dst.copy_(torch.as_strided(qb.data, (3 * ch,), (1,))))
            elif isinstance(mod, _Sampler):
                ch = mod.conv.weight.shape[0]
                job(w.samplers[name].w, mod.conv.weight, ch, ch, 3)
                job(tw.samplers[name].wd, mod.conv.weight, ch, ch, 3, dgrad=1)
                if ".upsamplers." in name:      # the sub-pixel phase kernels: pre-summed taps first (one contraction), then packed like any weight
                    src4, wt = w.samplers[name].w4_src, mod.conv.weight
                    self.pre.append(lambda src4=src4, wt=wt: upsample_phase_weights_stacked(wt.data, out=src4))
                    for p in range(4):
                        job(w.samplers[name].w4[p], src4[p], ch, ch, 2)
                        job(tw.samplers[name].wd4[p], src4[p], ch, ch, 2, dgrad=1)
        co, c0 = m.conv_out.weight.shape[0], m.conv_out.weight.shape[1]
        job(w.conv_out_w, m.conv_out.weight, co, c0, 3, cout_pad=w.conv_out_pad)
        job(tw.conv_out_d, m.conv_out.weight, c0, co, 3, dgrad=1, cin_pad=w.conv_out_pad)
        job(tw.conv_in_d, m.conv_in.weight, ci, m.conv_in.weight.shape[0], 3, dgrad=1, cout_pad=32)
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
        if getattr(w, "class_mode", None) == "timestep":      # the class MLP's transposed weights (its biases alias the parameters)
            ce = m.class_embedding
            self.small += [lambda: w.cw1T.copy_(ce.linear_1.weight.data.t()), lambda: w.cw2T.copy_(ce.linear_2.weight.data.t())]
        for t in (w.b1, w.b2, w.conv_in_b):
            pass  # alias the master parameters (fp32, contiguous, same device): nothing to refresh
        self._alias_check = [(w.b1, te.linear_1.bias), (w.b2, te.linear_2.bias), (w.conv_in_b, m.conv_in.bias),
                             (w.gn_out[0], m.conv_norm_out.weight)]
        for a, b in self._alias_check:
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


class UNetTrainer:
    """One optimisation step of ``perform_training_epoch`` (``utils_training.py:244-454``) on the HIP engine:
    forward -> loss (+ d loss / d out) -> backward -> [gradient all-reduce] -> clip + AdamW + EMA -> re-pack weights."""

    def __init__(self, model: CustomCondUNet2DModel, scheduler, lr: float, *, device=None, use_ema: bool = True,
                 max_grad_norm: Optional[float] = 1.0, group=None, trainable=None, **adamw):
        """``trainable``: which parameters train -- names, a predicate ``f(name, parameter)``, or None = what the parameters'
        ``requires_grad`` flags say (:func:`phendiff_amd.training.resolve_trainable`: the reference freezes with
        ``requires_grad_(False)`` and ``--attention_fine_tuning`` re-enables ``module.attentions``, train.py:189-220).  Frozen
        parameters get no gradient launches, no AdamW / EMA update, and stay out of the global norm and of the all-reduce buckets."""
        from .training import DiffusionLoss, FlatAdamWEMA, broadcast_from_rank0_, resolve_trainable
        self.model, self.scheduler = model, scheduler
        dev = device or model.device
        if torch.device(dev).type != "cuda":
            raise L.PhenDiffHipError("phendiff_amd trains on MI355X only (no CPU fallback): move the model to 'cuda'")
        order = training_param_order(model)
        flags = resolve_trainable(order, model, trainable)
        if not any(flags):
            raise ValueError("UNetTrainer: no trainable parameter (every parameter is frozen)")
        self.frozen = frozenset(n for (n, _), f in zip(order, flags) if not f)
        self.opt = FlatAdamWEMA([p for _, p in order], lr, use_ema=use_ema, max_grad_norm=max_grad_norm, **adamw)
        self.opt.set_trainable(flags)
        tail = [(n, p, f) for (n, p), f in zip(order, flags) if n.startswith("class_embedding.")]
        if tail and all(f for _, _, f in tail):
            self.opt.set_tail(sum(p.numel() for _, p, _ in tail), tuple(n for n, _, _ in tail))
        self._cond = True
        # DDP's wrap-time broadcast (train.py:311-326): rank 0's parameters everywhere; the EMA shadow starts from them
        broadcast_from_rank0_(self.opt.flat, group)
        if self.opt.ema is not None:
            self.opt.ema.copy_(self.opt.flat)
        self.params = {n: p.data for n, p in order}
        self.grads = {n: p.grad for n, p in order}
        model.invalidate()
        self.loss_fn = DiffusionLoss(scheduler, dev)
        # --mixed_precision fp16 (args_parser.py:381-390): fp16 activation gradients need the loss scale accelerate's GradScaler applies
        if getattr(model, "compute_dtype", None) == "fp16":
            from .training import LossScaler
            self.opt.scaler = LossScaler()
        self.device = dev
        self._plans = {}
        self._tw = None
        self._repack = None

    # kernel-layout weight copies: the model owns the inference set (``model._weights``), the trainer the gradient set and the
    # re-packer.  ``model.to()`` / ``.cuda()`` / ``load_state_dict()`` / ``pipeline.to()`` drop ``model._weights`` (``invalidate``)
    # and a later inference forward rebuilds it as a NEW object: everything the trainer cached is bound to the old one.
    def _make_packed(self):
        return _PackedWeights(self.model, self.device)

    def _make_train_weights(self):
        return TrainWeights(self.model, self.device, self.model._weights.tdt)

    def _make_repacker(self):
        return _Repacker(self.model, self.model._weights, self._tw)

    def _make_plan(self, key):
        m = self.model
        return UNetTrainPlan(m, m._weights, self._tw, *key, self.device, self.params, self.grads, frozen=self.frozen)

    def _bind_weights(self) -> bool:
        """Make the trainer's caches refer to the model's CURRENT packed weights; returns True when they had to be rebuilt
        (plans dropped, gradient-layout weights and re-packer rebuilt)."""
        m = self.model
        if m._weights is not None and m._weights is getattr(self, "_bound_w", None):
            return False
        for n, p in m.named_parameters():
            t = self.params.get(n)
            if t is not None and p.data_ptr() != t.data_ptr():
                raise L.PhenDiffHipError(f"parameter {n} no longer aliases the trainer's flat buffer (the model was converted or "
                                         "moved after the trainer was built): build a new trainer")
        if m._weights is None:
            m._weights = self._make_packed()
        self._tw = self._make_train_weights()
        # ONE gradient-layout set per model: the model's input-gradient plans (gradient-guided transfer of a model fine-tuned
        # in the same process, utils_Img2Img.py:651-760 after train.py) read the trainer's set, which the re-packer refreshes after
        # every optimizer step; input-gradient plans built on an older set are dropped
        m._grad_weights = self._tw
        for k in [k for k in m._plans if isinstance(k, tuple) and k and k[0] == "input_grad"]:
            del m._plans[k]
        self._repack = None
        self._plans = {}
        self._bucket_key = None
        self._bound_w = m._weights
        return True

    def plan_for(self, B, H, W):
        self._bind_weights()
        key = (B, H, W)
        p = self._plans.get(key)
        if p is None:
            p = self._make_plan(key)
            self._plans[key] = p
        return p

    def forward_backward(self, noisy, timesteps, clean, noise, class_labels=None, class_emb=None, after_op=None):
        """Loss of one batch and its parameter gradients (accumulated into the flat gradient buffer).  ``after_op``: hooks the
        overlapped data-parallel path hangs on backward launches (gradient buckets becoming final)."""
        B, _, H, W = noisy.shape
        plan = self.plan_for(B, H, W)
        st = torch.cuda.current_stream(self.device).cuda_stream
        x = noisy.contiguous().float()
        ts = timesteps.to(device=self.device, dtype=torch.float32).contiguous()
        if getattr(plan.w, "class_mode", None) == "identity" and class_emb is None and class_labels is not None:
            class_emb, class_labels = class_labels, None          # class_embed_type = "identity": the "labels" are the embedding rows
        labels = class_labels.to(device=self.device, dtype=torch.int64).contiguous() if class_labels is not None else None
        cemb = class_emb.to(device=self.device, dtype=torch.float32).contiguous() if class_emb is not None else None
        self._cond = labels is not None        # the class table has a gradient only when the labels went through it
        out = torch.empty_like(x)
        plan.forward(x, ts, labels, cemb, out, st)
        loss, dout = self.loss_fn(out, clean, noise, timesteps, grad_scale=self.opt.scaler.scale if self.opt.scaler is not None else 1.0)
        plan.backward(dout, st, after_op=after_op)
        return loss, out

    def step(self, noisy, timesteps, clean, noise, class_labels=None, class_emb=None, lr: Optional[float] = None, group=None,
             overlap: bool = True, bucket_bytes: int = 16 << 20):
        """forward -> loss -> backward, with the data-parallel gradient all-reduce (RCCL; what DDP does under
        ``accelerator.backward``, train.py:311-326) running bucket by bucket on a second stream WHILE the backward still
        computes the earlier layers' gradients -> clip + AdamW + EMA -> re-pack."""
        import torch.distributed as dist
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        if world == 1 and not getattr(self, "force_collectives", False):
            loss, _ = self.forward_backward(noisy, timesteps, clean, noise, class_labels, class_emb)
        elif not overlap:
            from .training import allreduce_mean_ranges_
            loss, _ = self.forward_backward(noisy, timesteps, clean, noise, class_labels, class_emb)
            allreduce_mean_ranges_(self.opt.grad, self.opt.trainable_ranges(), group)
        else:
            loss = self._forward_backward_overlapped(noisy, timesteps, clean, noise, class_labels, class_emb, group, world,
                                                     bucket_bytes)
        self._optimizer_step(lr)
        self.refresh_weights()
        return loss

    def _optimizer_step(self, lr):
        # unconditional step (class_emb = zeros, utils_training.py:510-516): the class table's .grad stays None in the
        # reference, torch's AdamW skips it (no decay, no moment update, no step count) -- EMA still steps
        self.opt.step(lr, tail_active=self._cond)

    def _forward_backward_overlapped(self, noisy, timesteps, clean, noise, class_labels, class_emb, group, world, bucket_bytes):
        import torch.distributed as dist
        B, _, H, W = noisy.shape
        plan = self.plan_for(B, H, W)
        key = (id(plan), bucket_bytes)
        if getattr(self, "_bucket_key", None) != key:
            names = list(self.grads)
            last = len(plan.bwd_ops) - 1
            # buckets are cut inside each run of trainable parameters: a frozen parameter's gradient segment is not exchanged
            self._buckets, off, run = [], 0, [0, [], []]
            frozen = getattr(self, "frozen", frozenset())
            for n in names + [None]:
                if n is None or n in frozen:
                    if run[1]:
                        self._buckets += [(run[0] + a, run[0] + b, r) for a, b, r in plan_grad_buckets(run[1], run[2], max(1, bucket_bytes // 4))]
                    if n is None:
                        break
                    off += self.grads[n].numel()
                    run = [off, [], []]
                    continue
                k = self.grads[n].numel()
                run[1].append(k)
                run[2].append(plan.grad_ready.get(n, last))
                off += k
            self._buckets.sort(key=lambda b: b[2])
            self._bucket_key = key
            self._comm_stream = torch.cuda.Stream(device=self.device)
        cur = torch.cuda.current_stream(self.device)
        comm = self._comm_stream
        flat = self.opt.grad
        works, hooks = [], {}

        native = getattr(self, "native_comm", None)      # phendiff_amd.comm.NativeComm: RCCL through the C ABI (pd_allreduce_bucket)

        timing = getattr(self, "comm_timing", False)      # bench.py: events around every bucket's collective on the comm stream
        skip = getattr(self, "skip_collectives", False)   # bench.py: the same step with the exchange left out (`step_ms_no_comm`)
        if timing:
            self._comm_events = []

        def launch(start, end):
            ev = torch.cuda.Event()
            ev.record(cur)
            comm.wait_event(ev)
            if skip:
                return
            if timing:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(comm)
            if native is not None:
                native.allreduce_(flat[start:end], mean=False, algo=1, stream=comm)
            else:
                with torch.cuda.stream(comm):
                    wk = dist.all_reduce(flat[start:end], op=dist.ReduceOp.SUM, group=group, async_op=True)
                    if timing:
                        wk.wait()                  # the COMM stream waits for the collective (RCCL runs it on a stream of its own)
                    else:
                        works.append(wk)
            if timing:
                e1.record(comm)
                self._comm_events.append((e0, e1, (end - start) * 4))

        for start, end, rdy in self._buckets:
            prev = hooks.get(rdy)
            hooks[rdy] = (lambda s=start, e=end, p=prev: ((p() if p else None), launch(s, e)))
        loss, _ = self.forward_backward(noisy, timesteps, clean, noise, class_labels, class_emb, after_op=hooks)
        if timing:
            self._bwd_end_event = torch.cuda.Event(enable_timing=True)
            self._bwd_end_event.record(cur)
        for wk in works:
            wk.wait()                      # the compute stream waits for the collectives, the host does not
        cur.wait_stream(comm)
        flat.div_(world)
        return loss

    def comm_overlap_stats(self):
        """After a step run with ``comm_timing = True`` (and a device synchronisation): how much of the gradient exchange ran under the
        backward.  comm_ms = busy time of the comm stream (sum over the buckets' collectives, which run one after the other on it);
        exposed_ms = what was left of it when the backward's last launch finished (the optimizer waits that long); overlap_frac =
        (comm_ms - exposed_ms) / comm_ms -- SURVEY 8(d) cfg4's "all-reduce overlap fraction"."""
        evs = getattr(self, "_comm_events", None)
        if not evs:
            return None
        comm_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in evs)
        exposed = max(0.0, self._bwd_end_event.elapsed_time(evs[-1][1]))
        exposed = min(exposed, comm_ms)
        nbytes = sum(b for _, _, b in evs)
        return {"comm_ms": round(comm_ms, 3), "exposed_ms": round(exposed, 3),
                "overlap_frac": round((comm_ms - exposed) / comm_ms, 4) if comm_ms > 0 else None,
                "buckets": len(evs), "bytes": nbytes,
                "algbw_GBs": round(nbytes / (comm_ms * 1e-3) / 1e9, 2) if comm_ms > 0 else None}

    def use_native_comm(self, comm=None, group=None):
        """Route the overlapped gradient exchange through ``pd_allreduce_bucket`` (reduce-scatter + all-gather on RCCL behind the C
        ABI) instead of ``torch.distributed.all_reduce``.  ``comm``: a ``NativeComm``; default: one built over the process group."""
        from .comm import NativeComm
        self.native_comm = comm if comm is not None else NativeComm.from_process_group(group, self.device)
        return self.native_comm

    def save_state(self, output_dir, **kw):
        """``accelerator.save_state`` layout (``train_state.py``): model, AdamW moments, LR-scheduler, RNG, EMA."""
        from .train_state import save_state
        save_state(self, output_dir, **kw)

    def load_state(self, input_dir, **kw):
        """``accelerator.load_state``: restores everything ``save_state`` wrote, in place (plans stay valid)."""
        from .train_state import load_state
        return load_state(self, input_dir, **kw)

    def refresh_weights(self):
        """Parameters changed in place: rebuild the kernel-layout copies (same device buffers, plans stay valid).  If the model
        dropped or replaced its packed weights since (``invalidate``), rebind first -- the replacement may have been packed
        from older parameters, so it is re-packed as well."""
        m = self.model
        if m._weights is None and getattr(self, "_bound_w", None) is None:
            return                                   # nothing packed yet: the first plan packs the current parameters
        self._bind_weights()
        if self._repack is None:
            self._repack = self._make_repacker()
        self._repack.run(torch.cuda.current_stream(self.device).cuda_stream)
