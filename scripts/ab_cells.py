#!/usr/bin/env python3
"""Whole-workload A/B of kernel-selecting switches INSIDE ONE PROCESS (round 5).

    python scripts/ab_cells.py [--rounds 4] [--steps 10] [--batch 32] [--out FILE] CELL CELL ...

A CELL is a comma-separated list of VAR=value words, or "-" for the library's defaults.  For every cell the headline workload
(bench.py: 256x256, S = 50 + 50, super_small, bf16, B = 32) is captured into its OWN hipGraph on its own launch plan while the
cell's variables are in the environment -- the library reads its diagnostic switches at every dispatch (pd_common.h diag_env) and
unet.py reads PD_SUBPIXEL_UP when a plan is built, so the captured graphs differ exactly by the kernels those switches select.
Then ROUNDS rounds replay the cells in alternating order, STEPS trajectories each, timed like bench.py's region (device-resident
batch -> pinned host array, synchronised at both ends).  One process, one box, one thermal history: the cells see the same chip
in the same state, and a round's cells are seconds apart.  Between two cells one untimed replay of the next cell settles the
clocks on ITS instruction mix (the workload is power-coupled: DESIGN section 6).

`--smi` samples `rocm-smi` (sclk, power) during each timed block in a side thread (a child process per sample; nothing in
this process execs).
"""
import argparse
import collections
import json
import os
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def smi_sample(raw=False):
    """(sclk MHz, socket power W) of card 0 from one `rocm-smi` child process (key names differ between releases: matched loosely)."""
    import re
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=10)
        if raw:
            return r.stdout[:1500]
        j = json.loads(r.stdout[r.stdout.index("{"):])
        card = j[sorted(k for k in j if k.startswith("card"))[0]]
        sclk = next((v for k, v in card.items() if "sclk" in k.lower()), None)
        pw = next((v for k, v in card.items() if "power" in k.lower() and "(w)" in k.lower()), None)
        num = lambda v: float(re.search(r"[-+]?\d+(\.\d+)?", str(v)).group(0))      # noqa: E731
        return (num(sclk) if sclk else None), (num(pw) if pw else None)
    except Exception:      # noqa: BLE001
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cells", nargs="+")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--inference-steps", type=int, default=50)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--smi", action="store_true")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    sys.path.insert(0, ROOT)
    import bench
    import phendiff_amd as P
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B, S, size = args.batch, args.inference_steps, args.size
    x, labels = bench.synth_batch(B, size, 1234)
    x, labels = x.to(dev), labels.to(dev)
    target = P.swap_binary_labels(labels)
    host_out = torch.empty((B, size, size, 3), dtype=torch.float32, pin_memory=True)

    cells = collections.OrderedDict()
    touched = set()
    for cell in args.cells:
        env = {} if cell == "-" else dict(w.split("=", 1) for w in cell.split(","))
        touched |= set(env)
    for cell in args.cells:
        env = {} if cell == "-" else dict(w.split("=", 1) for w in cell.split(","))
        for k in touched:
            os.environ.pop(k, None)
        os.environ.update(env)
        # a model of its own per cell: packed weights (the q/k/v fold, the sub-pixel phase kernels) are part of what a switch selects
        torch.manual_seed(0)
        unet = P.CustomCondUNet2DModel(compute_dtype=args.dtype, **dict(P.UNET_CONFIGS["super_small"], sample_size=size))
        pipe = P.ConditionalDDIMPipeline(unet.to(dev), P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"]))
        runner = P.DDIBGraph(pipe, batch_size=B, num_inference_steps=S)
        kinds = collections.Counter(op.what for op in runner.plan.ops)
        cells[cell] = (runner, pipe)
        print(f"captured {cell!r}: {len(runner.plan.ops)} launches per forward {dict(kinds)}", flush=True)
    for k in touched:
        os.environ.pop(k, None)

    def one_batch(runner):
        runner.run(x, labels, target)
        host_out.copy_(runner.images, non_blocking=True)

    outputs = {}
    for cell, (runner, _) in cells.items():      # first replay of every graph (upload, first touch) + the result, for the record
        one_batch(runner)
        torch.cuda.synchronize(dev)
        outputs[cell] = host_out.clone()
    ref = outputs[args.cells[0]]
    for cell, o in outputs.items():
        print(f"output of {cell!r} vs the first cell: max |diff| = {float((o - ref).abs().max()):.3e}", flush=True)

    if args.smi:
        print("rocm-smi sample:", smi_sample(), "raw:", smi_sample(raw=True).replace("\n", " ")[:600], flush=True)
    results = collections.defaultdict(list)
    smi = collections.defaultdict(list)
    in_situ = {}
    order = list(cells)
    for r in range(args.rounds):
        for cell in (order if r % 2 == 0 else order[::-1]):
            runner = cells[cell][0]
            one_batch(runner)                     # settle on this cell's instruction mix (untimed)
            torch.cuda.synchronize(dev)
            stop = threading.Event()
            samples = []

            def sampler():
                while not stop.wait(2.0):
                    samples.append(smi_sample())
            th = threading.Thread(target=sampler, daemon=True) if args.smi else None
            if th:
                th.start()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                one_batch(runner)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            stop.set()
            if th:
                th.join()
                smi[cell] += [s for s in samples if s[0]]
            v = B * args.steps / dt
            results[cell].append(v)
            extra = ""
            if samples and any(s[0] for s in samples):
                ok = [s for s in samples if s[0]]
                extra = f"  sclk {statistics.median(s[0] for s in ok):.0f} MHz  power {statistics.median(s[1] for s in ok if s[1]):.0f} W"
            print(f"round {r + 1}  {cell:58s} {v:8.4f} images/s{extra}", flush=True)
    # the dominant kernel's launch time IN SITU under every cell (bench._in_situ_launch_ms: steady-state eager replay of the cell's plan)
    for cell in order:
        runner = cells[cell][0]
        one_batch(runner)
        for k in touched:                         # the eager replay dispatches anew: the cell's switches go back into the environment
            os.environ.pop(k, None)
        os.environ.update({} if cell == "-" else dict(w.split("=", 1) for w in cell.split(",")))
        st = torch.cuda.current_stream(dev).cuda_stream
        ms, _, n = bench._in_situ_launch_ms(runner.plan, "attn_d8", st)
        torch.cuda.synchronize(dev)
        in_situ[cell] = ms
        print(f"in situ  {cell:58s} attn_d8 {ms:.4f} ms per launch ({n} samples)", flush=True)
    base = statistics.median(results[order[0]])
    lines = []
    for cell in order:
        v = results[cell]
        m = statistics.median(v)
        s = smi.get(cell) or []
        extra = f"  sclk {statistics.median(a for a, _ in s):.0f} MHz  power {statistics.median(b for _, b in s if b):.0f} W" if s else ""
        lines.append(f"{cell:58s} median {m:8.4f}  x{m / base:.4f}  min {min(v):.4f} max {max(v):.4f}  attn in situ {in_situ[cell]:.4f} ms{extra}  {[round(a, 3) for a in v]}")
    print("\n".join(lines), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"cells": {c: results[c] for c in order}, "rounds": args.rounds, "steps": args.steps, "batch": B,
                       "smi": {c: smi.get(c) for c in order}, "attn_d8_in_situ_ms": in_situ}, f)


if __name__ == "__main__":
    main()
