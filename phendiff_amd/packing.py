"""Host-side weight packing into the MFMA fragment order ``pd_conv`` consumes
(``include/phendiff_hip.h``: ``pd_conv_args.w_packed``)."""
from __future__ import annotations

import torch


def pack_conv_weight(w: torch.Tensor, dtype: torch.dtype, cout_pad: int | None = None) -> torch.Tensor:
    """OIHW conv weight (or ``[out, in, 1, 1]`` linear) -> ``[Cout_pad/32][Cin/32][taps][2][64][8]``.

    For lane ``l`` (``r = l & 31``, ``h = l >> 5``) of fragment ``(ct, chunk, tap, s)`` element ``j`` is
    ``W[32*ct + r][32*chunk + 16*s + 8*h + j][tap]`` -- i.e. each wave reads one fragment as 1 KiB (bf16) /
    2 KiB (fp32) of contiguous memory, already in the A-operand layout of ``v_mfma_f32_32x32x16_bf16``
    (and, with the k order shared by A and B, of the 8 x ``v_mfma_f32_32x32x2_f32`` fp32 form).
    """
    assert w.ndim == 4
    cout, cin, kh, kw = w.shape          # rectangular kernels (pd_conv_rect: 1x7, 7x1, 1x3, 3x1): tap = ky * KW + kx
    assert cin % 32 == 0, "input channels must be a multiple of 32"
    cp = cout_pad or cout
    assert cp % 32 == 0 and cp >= cout
    taps = kh * kw
    wf = torch.zeros((cp, cin, taps), dtype=torch.float32, device=w.device)
    wf[:cout] = w.reshape(cout, cin, taps).float()
    # [ct, r, chunk, s, h, j, tap] -> [ct, chunk, tap, s, h, r, j]
    wf = wf.reshape(cp // 32, 32, cin // 32, 2, 2, 8, taps).permute(0, 2, 6, 3, 4, 1, 5).contiguous()
    return wf.reshape(cp // 32, cin // 32, taps, 2, 64, 8).to(dtype)


def dgrad_weight(w: torch.Tensor) -> torch.Tensor:
    """Weight of the input-gradient convolution: for y = conv(x, W) (stride 1, "same" padding),
    dx = conv(dy, W') with W'[ci, co, ky, kx] = W[co, ci, K-1-ky, K-1-kx]  (transpose + spatial flip)."""
    return w.permute(1, 0, 2, 3).flip(2, 3).contiguous()


def upsample_phase_weights_stacked(w: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """``[4][Cout][Cin][2][2]`` fp32: the four phase kernels of :func:`upsample_phase_weights` stacked (phase 2 a + b), formed by ONE
    contraction on the weight's device -- R_a w R_b^T with R_0 = [[1,0,0],[0,1,1]] (taps {0 | 1+2}), R_1 = [[1,1,0],[0,0,1]] ({0+1 | 2}) --
    so that the training re-pack can refresh them every step (``out``: a persistent buffer the pack jobs read)."""
    assert w.ndim == 4 and w.shape[2:] == (3, 3)
    if w.is_cuda:
        # on the device (model load, and after every optimizer step of a fine-tuning run): one launch of the HIP library, the same sums in the
        # same order (rows, then columns) -- torch.einsum lowered this to two fp32 GEMMs + copies, 1.8 ms of every SD-2.1 step
        import ctypes as C
        from . import _lib as L
        wf = w.detach()
        if wf.dtype != torch.float32 or not wf.is_contiguous():
            wf = wf.float().contiguous()
        if out is None:
            out = torch.empty((4, w.shape[0], w.shape[1], 2, 2), dtype=torch.float32, device=w.device)
        assert out.is_contiguous() and out.dtype == torch.float32 and out.shape == (4, w.shape[0], w.shape[1], 2, 2)
        a = L.UpsamplePhaseWeightsArgs(cout=w.shape[0], cin=w.shape[1], w=wf.data_ptr(), out=out.data_ptr())
        with torch.cuda.device(w.device):
            L.check(L.lib().pd_upsample_phase_weights(C.byref(a), torch.cuda.current_stream().cuda_stream), "pd_upsample_phase_weights")
        return out
    R = torch.tensor([[[1., 0., 0.], [0., 1., 1.]], [[1., 1., 0.], [0., 0., 1.]]], dtype=torch.float32, device=w.device)
    k = torch.einsum("auy,oiyx,bvx->aboiuv", R, w.detach().float(), R).reshape(4, w.shape[0], w.shape[1], 2, 2)
    if out is not None:
        out.copy_(k)
        return out
    return k.contiguous()


def upsample_phase_weights(w: torch.Tensor):
    """The sub-pixel form of ``F.interpolate(x, scale_factor=2, mode="nearest")`` followed by a 3x3 pad-1 convolution (diffusers
    ``Upsample2D``): output pixel (2y + a, 2x + b) only sees the 2x2 block of LOW-resolution pixels rows y - (1 - a) .. +1,
    columns x - (1 - b) .. +1, each through the sum of the 3x3 taps that land on it.  Returns the four 2x2 kernels
    ``[W_00, W_01, W_10, W_11]`` (``W_ab``: OI22, fp32) for ``pd_conv(phase = 1 + 2 a + b)`` -- 4 / 9 of the multiply-adds of the
    convolution over the upsampled tensor, the same function of the weights (sums formed in fp32 before the 16-bit rounding)."""
    return list(upsample_phase_weights_stacked(w))
