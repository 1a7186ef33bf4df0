"""Per-parameter gradient error of the HIP UNet backward against torch.autograd over the CPU oracle (diagnostic checker: lives
under tests/ because only the tests, smoke() and the bench CPU baseline may use the oracle).  python tests/check_backward.py"""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phendiff_amd as P  # noqa: E402
from phendiff_amd.unet_train import UNetTrainer  # noqa: E402
from oracle import CondUNet2DRef  # noqa: E402


def main(mode="f32", size=32, B=2, name="super_small"):
    torch.manual_seed(0)
    keys = CondUNet2DRef.__init__.__code__.co_varnames
    r = CondUNet2DRef(**{k: v for k, v in dict(P.UNET_CONFIGS[name], sample_size=size).items() if k in keys})
    m = P.CustomCondUNet2DModel(compute_dtype=mode, **dict(P.UNET_CONFIGS[name], sample_size=size))
    m.load_state_dict(r.state_dict())
    m = m.to("cuda:0")
    sched = P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"])
    g = torch.Generator().manual_seed(5)
    clean = torch.rand(B, 3, size, size, generator=g) * 2 - 1
    noise = torch.randn(B, 3, size, size, generator=g)
    ts = torch.tensor([2500, 700, 40, 1500][:B])
    labels = torch.arange(B) % 2
    acp = sched.alphas_cumprod[ts]
    sa, sb = (acp ** 0.5).view(-1, 1, 1, 1), ((1 - acp) ** 0.5).view(-1, 1, 1, 1)
    noisy = sa * clean + sb * noise
    for p in r.parameters():
        p.requires_grad_(True)
    out = r(noisy, ts, class_labels=labels).sample
    target = sa * noise - sb * clean
    loss = torch.nn.functional.mse_loss(out, target)
    loss.backward()
    ref = {n: p.grad for n, p in r.named_parameters()}

    tr = UNetTrainer(m, sched, lr=1e-4, use_ema=False)
    l2, out2 = tr.forward_backward(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda())
    torch.cuda.synchronize()
    print("loss", float(loss), float(l2), "out rel", float((out2.cpu() - out.detach()).norm() / out.detach().norm()))
    worst = []
    num = den = 0.0
    for n, gr in ref.items():
        got = tr.grads[n].cpu()
        e = float((got - gr).norm() / (gr.norm() + 1e-30))
        num += float((got - gr).double().pow(2).sum()); den += float(gr.double().pow(2).sum())
        worst.append((e, n, float(gr.norm())))
    if os.environ.get("PD_ALL"):
        for e, n, nn in worst:
            print(f"{e:10.3e}  |g|={nn:9.3e}  {n}")
    worst.sort(reverse=True)
    for e, n, nn in worst[:25]:
        print(f"{e:10.3e}  |g|={nn:9.3e}  {n}")
    print("global rel", (num / den) ** 0.5, "n params", len(worst))


if __name__ == "__main__":
    main(*(sys.argv[1:2] or ["f32"]), size=int(sys.argv[2]) if len(sys.argv) > 2 else 32)
