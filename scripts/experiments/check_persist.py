#!/usr/bin/env python3
"""Persistent vs one-tile-per-workgroup pd_conv on the same inputs (multi-item workgroups, which the small kernel tests never
reach): run in two processes (PD_NO_PERSIST_CONV=1 / unset) and compare.  GPU only.
    python scripts/check_persist.py            # drives both runs and compares"""
import ctypes as C, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [  # B, H, C0, C1, Cout, gn, res, up, tail(C0,C1), dtype
    (8, 256, 64, 0, 64, 1, 1, 0, None, "bf16"),
    (8, 128, 128, 64, 128, 1, 0, 0, None, "bf16"),
    (8, 64, 256, 0, 256, 0, 0, 1, None, "bf16"),
    (6, 128, 64, 0, 64, 1, 0, 0, (128, 64), "bf16"),
    (3, 72, 64, 0, 96, 1, 1, 0, None, "bf16"),          # partial tiles, Cout not a multiple of 64
    (4, 64, 64, 32, 64, 1, 1, 0, (64, 0), "f32"),
]


def run(tag):
    import torch
    from phendiff_amd import _lib as L
    from phendiff_amd.packing import pack_conv_weight
    lib = L.lib(); dev = "cuda:0"; out = {}
    for ci, (B, H, c0, c1, cout, gn, res, up, tail, dt) in enumerate(CASES):
        torch.manual_seed(ci)
        code, tdt = (1, torch.bfloat16) if dt == "bf16" else (0, torch.float32)
        x0 = torch.randn(B, H, H, c0, device=dev).to(tdt)
        x1 = torch.randn(B, H, H, c1, device=dev).to(tdt) if c1 else None
        cin = c0 + c1
        hc = 2 * H if up else H
        tc = (tail[0] + tail[1]) if tail else 0
        w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
        if tail:
            w2, ws = pack_conv_weight(w, tdt), pack_conv_weight(torch.randn(cout, tc, 1, 1) / tc ** 0.5, tdt)
            ct = w2.shape[0]      # per 32-co tile the tail's fragments follow the 3x3 fragments (unet.py _PackedWeights)
            wp = torch.cat([w2.reshape(ct, -1, 64, 8), ws.reshape(ct, -1, 64, 8)], 1).contiguous().to(dev)
        else:
            wp = pack_conv_weight(w, tdt).to(dev)
        t0 = torch.randn(B, hc, hc, tail[0], device=dev).to(tdt) if tail else None
        t1 = torch.randn(B, hc, hc, tail[1], device=dev).to(tdt) if (tail and tail[1]) else None
        bias = torch.randn(cout, device=dev)
        temb = torch.randn(B, 512, device=dev)
        y = torch.zeros(B, hc, hc, cout, device=dev, dtype=tdt)
        r = torch.randn(B, hc, hc, cout, device=dev).to(tdt) if res else None
        sc = (torch.rand(B, cin, device=dev) + 0.5) if gn else None
        sh = torch.randn(B, cin, device=dev) if gn else None
        T = lib.pd_conv_stat_tiles(hc, hc, 3, 1)
        st = torch.zeros(B, T, cout, 2, device=dev)
        a = L.ConvArgs(dtype=code, B=B, Hin=H, Win=H, Hout=hc, Wout=hc, C0=c0, C1=c1, Cout=cout, Cout_pad=-(-cout // 32) * 32, ksize=3, stride=1,
                       pad=1, upsample=up, silu=gn, out_mode=0, heads=0, x0=x0.data_ptr(), x1=L.ptr(x1), scale=L.ptr(sc), shift=L.ptr(sh),
                       w_packed=wp.data_ptr(), bias=bias.data_ptr(), temb=temb.data_ptr(), temb_stride=512, residual=L.ptr(r),
                       y=y.data_ptr(), stats_out=st.data_ptr(), tail_x0=L.ptr(t0), tail_x1=L.ptr(t1), tail_C0=tail[0] if tail else 0,
                       tail_C1=tail[1] if tail else 0)
        L.check(lib.pd_conv(C.byref(a), torch.cuda.current_stream().cuda_stream), "pd_conv")
        torch.cuda.synchronize()
        out[f"y{ci}"] = y.float().cpu(); out[f"s{ci}"] = st.sum(1).cpu()
    torch.save(out, f"/tmp/check_persist_{tag}.pt")


if len(sys.argv) > 1:
    run(sys.argv[1])
else:
    import torch
    env = dict(os.environ)
    subprocess.run([sys.executable, __file__, "new"], check=True, env=env)
    subprocess.run([sys.executable, __file__, "old"], check=True, env=dict(env, PD_NO_PERSIST_CONV="1"))
    a, b = torch.load("/tmp/check_persist_new.pt"), torch.load("/tmp/check_persist_old.pt")
    bad = 0
    for k in a:
        d = (a[k] - b[k]).abs().max().item(); ref = b[k].abs().max().item()
        ok = d <= (2e-2 if k[0] == "y" else 2e-3) * max(ref, 1.0)
        bad += not ok
        print(f"{k}: max|new-old| {d:.3e} (max|old| {ref:.3e}) {'ok' if ok else 'MISMATCH'}")
    sys.exit(1 if bad else 0)
