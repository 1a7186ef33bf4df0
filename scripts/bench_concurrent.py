#!/usr/bin/env python3
"""Two DDIB trajectories replayed CONCURRENTLY on separate streams, the second delayed by a fraction of one UNet forward so that
its MFMA-bound conv sections run next to the other's VALU-bound attention section.  GPU only (diagnostic)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phendiff_amd as P
from phendiff_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16, help="images per trajectory")
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--delays", default="0,0.25,0.5,0.75", help="delays of the second trajectory, in forwards")
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
unet = P.CustomCondUNet2DModel(compute_dtype="bf16", **dict(P.UNET_CONFIGS["super_small"], sample_size=a.size))
pipe = P.ConditionalDDIMPipeline(unet.to(dev), P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"]))
B = a.batch
runners = [P.DDIBGraph(pipe, batch_size=B, num_inference_steps=a.steps, private_plan=i > 0) for i in range(2)]
x = torch.rand(B, 3, a.size, a.size, device=dev) * 2 - 1
labels = torch.arange(B, device=dev) % 2
for r in runners:
    r.run(x, labels, 1 - labels)
torch.cuda.synchronize()
lib = L.lib()


def launch(r):
    L.check(lib.pd_graph_launch(r.graph, r.stream.cuda_stream), "pd_graph_launch")


# one trajectory alone
t0 = time.perf_counter(); launch(runners[0]); torch.cuda.synchronize(); t_one = time.perf_counter() - t0
fwd = t_one / (2 * a.steps)
print(f"one trajectory of B={B}: {t_one*1e3:.0f} ms ({B/t_one:.2f} img/s), forward+step {fwd*1e3:.2f} ms", flush=True)
# clock rate of torch.cuda._sleep: calibrate
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(100_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_s = 100_000_000 / (e0.elapsed_time(e1) * 1e-3)
for d in [float(v) for v in a.delays.split(",")]:
    best = 1e9
    for _ in range(a.reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        launch(runners[0])
        with torch.cuda.stream(runners[1].stream):
            if d > 0:
                torch.cuda._sleep(int(d * fwd * cyc_per_s))
        launch(runners[1])
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"two concurrent, delay {d:.2f} forward: {best*1e3:.0f} ms -> {2*B/best:.2f} img/s ({2*t_one/best:.3f}x of back-to-back)", flush=True)
