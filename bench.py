#!/usr/bin/env python3
"""bench.py -- img2img images/sec (50-step DDIM invert + 50-step denoise, 256x256), BASELINE.json's metric.

One "step" = one DDIB class transfer of one synthetic batch per GPU: `super_small` UNet at
sample_size 256, `3k_steps_clipping_rescaling` scheduler, S = 50 + 50, inputs resident in HBM when the timed
region starts, output = float NHWC [0,1] images in HBM (PNG encode / dataset decode excluded).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Images are independent: each rank transfers its own batches, no collective on the data path ("scaling": "weak").
Rank 0 prints ONE JSON line.  At N = 1 (the driver's default run) that line also carries `side_workloads`: short legs of
BASELINE.json's other configs -- configs[1] `train`, configs[3] `sd_train`, configs[4] `sd_img2img` in fp16 and bf16, and (round 6) the
reference's own precision for the two pixel configs: `img2img_fp16` (configs[2]) and `train_fp16` (configs[1]) -- each run as a
child process AFTER the headline's fields are final, each with its own value / ms_per_step / roofline / cpu_baseline
(`--no-side-workloads` skips them; `--workload X` runs one of them as the main workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# RCCL across processes needs dmabuf IPC on this driver stack (already exported on the pool's boxes; harmless otherwise)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

# SURVEY.md 8(d) / BASELINE.md 2: algorithmic work per UNet forward per image, super_small @256^2
FWD_GFLOP_PER_IMAGE = 376.0
FWD_ACT_MB_PER_IMAGE_BF16 = 691.8
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
PEAK_MFMA_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "f32": 157.3}   # dense peaks, MI355X_MICROARCH.md (F16 = BF16 rate)


def synth_batch(B, size, seed):
    """x ~ U[-1,1] with a class-dependent offset, labels = arange(B) % 2."""
    g = torch.Generator().manual_seed(seed)
    labels = torch.arange(B) % 2
    x = torch.rand(B, 3, size, size, generator=g) * 2 - 1
    x = (x + 0.25 * (2 * labels.float() - 1).view(B, 1, 1, 1)).clamp(-1, 1)
    return x, labels


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (os.cpu_count() reports the
    host's cores, which oversubscribes a quota-limited container badly)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:          # cgroup v2
            quota, period = f.read().split()
            if quota != "max":
                n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:   # v1
                quota, period = int(f.read()), int(g.read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, min(n, 64))


class SmiSampler:
    """Clock and power of card 0 under the workload (round 5): boxes of the pool differ by +-5.5 % on this power-coupled workload
    (15.3 - 17.1 images/s for identical sources), so the line says what the chip of THIS run held.  A side thread starts one short
    `rocm-smi` child process per period (nothing in this process execs; host-side only); failures are swallowed.  Round 6: never
    inside the timed region -- `bracket()` samples through a repeat of the timed steps right after it."""

    def __init__(self, period_s=2.0):
        import threading
        self.period, self.samples, self._stop = period_s, [], threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def sample():
        import re
        import subprocess
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=10)
            j = json.loads(r.stdout[r.stdout.index("{"):])
            card = j[sorted(k for k in j if k.startswith("card"))[0]]
            num = lambda v: float(re.search(r"[-+]?\d+(\.\d+)?", str(v)).group(0))      # noqa: E731
            sclk = next((num(v) for k, v in card.items() if "sclk" in k.lower()), None)
            power = next((num(v) for k, v in card.items() if "power" in k.lower() and "(w)" in k.lower()), None)
            return sclk, power
        except Exception:      # noqa: BLE001
            return None, None

    def _run(self):
        while not self._stop.wait(self.period):
            self.samples.append(self.sample())

    @classmethod
    def bracket(cls, run_steps):
        """no child process inside the timed region (the fork of a large GPU process every 2 s perturbs rank 0 only, and
        the reported time is the MAX over ranks).  The card's clock / power are read while it runs the SAME steps again right after
        the timed region (`run_steps()` queues them; the samples are taken while they execute), never while the clock is running."""
        s = cls(period_s=1.0).start()
        t0 = time.time()
        try:
            run_steps()
            while time.time() - t0 < 2.5:          # short steps (training): keep the card under the workload until a sample or two exist
                run_steps()
        finally:
            out = s.stop()
        if "source" in out:
            out["source"] = "rocm-smi --showclocks --showpower, every 1 s through a repeat of the timed steps AFTER the timed region (none inside it)"
        return out

    def start(self):
        self._th.start()
        return self

    def stop(self):
        self._stop.set()
        self._th.join(15)
        ok = [s for s in self.samples if s[0]]
        if not ok:
            return {"note": "rocm-smi gave no sample"}
        med = lambda v: sorted(v)[len(v) // 2]      # noqa: E731
        pw = [s[1] for s in ok if s[1]]
        return {"sclk_mhz_median": med([s[0] for s in ok]), "power_w_median": med(pw) if pw else None, "samples": len(ok),
                "source": "rocm-smi --showclocks --showpower, every 2 s through the timed region"}


def cpu_baseline(model_name, size, S, state_dict, seconds_budget=30.0):
    """The oracle (plain-PyTorch CPU fp32 restatement of the reference path) timed on this box's host cores on a
    bounded sample: B = 1 image, the first k inversion + first k denoising steps of the S-step schedules (per-step
    cost is step-independent), extrapolated to the 2*S-step trajectory."""
    import phendiff_amd as P
    from oracle import CondUNet2DRef, DDIMInverseSchedulerRef, DDIMSchedulerRef
    # thread count: the fastest of {8, 16, 32, 64, usable} on a small conv probe (a quota-limited container may report
    # far more cores than it can run; oversubscription costs >10x)
    import torch.nn.functional as F
    probe_x, probe_w = torch.randn(1, 64, 128, 128), torch.randn(64, 64, 3, 3)
    best = (float("inf"), 1)
    for nthr in sorted({c for c in (8, 16, 32, 64, usable_cores()) if c <= usable_cores()} or {1}):
        torch.set_num_threads(nthr)
        F.conv2d(probe_x, probe_w, padding=1)
        t0 = time.perf_counter()
        for _ in range(5):
            F.conv2d(probe_x, probe_w, padding=1)
        dt = time.perf_counter() - t0
        if dt < best[0] * 0.95:
            best = (dt, nthr)
    cores = best[1]
    torch.set_num_threads(cores)
    keys = CondUNet2DRef.__init__.__code__.co_varnames
    unet = CondUNet2DRef(**{k: v for k, v in dict(P.UNET_CONFIGS[model_name], sample_size=size).items() if k in keys}).eval()
    unet.load_state_dict(state_dict)
    fwd = DDIMSchedulerRef(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"])
    fwd.set_timesteps(S)
    inv = DDIMInverseSchedulerRef.from_config(fwd.config)
    inv.set_timesteps(S)
    x, labels = synth_batch(1, size, 1234)
    times = []
    t_start = time.time()
    with torch.no_grad():
        k = 0
        while k < S and (len(times) < 3 or time.time() - t_start < seconds_budget):
            for sched, ts, lab in ((inv, inv.timesteps, labels), (fwd, fwd.timesteps, 1 - labels)):
                t0 = time.perf_counter()
                out = unet(x, ts[k], class_labels=lab).sample
                sched.step(out, ts[k], x)
                times.append(time.perf_counter() - t0)
            k += 1
    if len(times) > 2:
        times = times[1:]            # the first step pages in the allocator / thread pool
    times.sort()
    t_step = times[len(times) // 2]
    return {"value": 1.0 / (2 * S * t_step), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle (CPU fp32 torch, {cores} threads), B=1 @{size}x{size}, median of {len(times)} UNet+scheduler "
                      f"steps ({k} inversion + {k} denoising of the S={S} schedules), extrapolated to {2 * S} steps; "
                      f"t_step={t_step:.3f}s"}


def cpu_baseline_train(model_name, size, state_dict, seconds_budget=30.0):
    """One optimisation step of the CPU oracle under torch.autograd + torch AdamW (what the reference's training loop runs)
    on a bounded batch."""
    from oracle import CondUNet2DRef
    import phendiff_amd as P
    threads = usable_cores()
    torch.set_num_threads(threads)
    keys = CondUNet2DRef.__init__.__code__.co_varnames
    r = CondUNet2DRef(**{k: v for k, v in dict(P.UNET_CONFIGS[model_name], sample_size=size).items() if k in keys})
    r.load_state_dict(state_dict)
    opt = torch.optim.AdamW(r.parameters(), lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    B = 2
    x, labels = synth_batch(B, size, 99)
    ts = torch.tensor([1500, 300])

    def step():
        out = r(x, ts, class_labels=labels).sample
        loss = torch.nn.functional.mse_loss(out, torch.zeros_like(out))
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(r.parameters(), 1.0)
        opt.step()
    step()
    t0, n = time.perf_counter(), 0
    while n < 1 or (time.perf_counter() - t0 < seconds_budget and n < 20):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": round(B / dt, 4), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{n} optimisation step(s) of the CPU oracle (torch.autograd + clip_grad_norm_ + AdamW) at batch {B}, "
                      f"{size}x{size}, fp32, {threads} threads"}


def _cpu_threads():
    import torch.nn.functional as F
    probe_x, probe_w = torch.randn(1, 64, 128, 128), torch.randn(64, 64, 3, 3)
    best = (float("inf"), 1)
    for nthr in sorted({c for c in (8, 16, 32, 64, usable_cores()) if c <= usable_cores()} or {1}):
        torch.set_num_threads(nthr)
        F.conv2d(probe_x, probe_w, padding=1)
        t0 = time.perf_counter()
        for _ in range(5):
            F.conv2d(probe_x, probe_w, padding=1)
        dt = time.perf_counter() - t0
        if dt < best[0] * 0.95:
            best = (dt, nthr)
    torch.set_num_threads(best[1])
    return best[1]


def cpu_baseline_sd_img2img(P, size, S):
    """The oracle's latent-diffusion transfer on this box's host cores, bounded: ONE image through the full-size stack (SD-2.1
    UNet 865.9 M + SD VAE, random init) -- one VAE encode, one inversion step, one denoising step, one VAE decode -- extrapolated
    to S + S steps (per-step cost is step-independent)."""
    from oracle import AutoencoderKLRef, CustomEmbeddingRef, DDIMSchedulerRef, SD21_UNET_CONFIG, SD_VAE_CONFIG, UNet2DConditionRef
    from oracle import class_emb_to_encoder_hidden_states as ehs_ref
    cores = _cpu_threads()
    torch.manual_seed(0)
    with torch.no_grad():
        unet = UNet2DConditionRef(**SD21_UNET_CONFIG).eval()
        vae = AutoencoderKLRef(**SD_VAE_CONFIG).eval()
        emb = CustomEmbeddingRef(2, 1024)
        x, labels = synth_batch(1, size, 1234)
        t0 = time.perf_counter()
        lat = vae.encode(x).latent_dist.sample() * 0.18215
        t_enc = time.perf_counter() - t0
        ehs = ehs_ref(emb(labels))
        sched = DDIMSchedulerRef(**P.SCHEDULER_CONFIGS["SD_orig_config"])
        sched.set_timesteps(S)
        ts = sched.timesteps
        steps = []
        for k in range(2):
            t0 = time.perf_counter()
            out = unet(lat, ts[k], ehs).sample
            lat = sched.step(out, ts[k], lat).prev_sample
            steps.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        vae.decode(lat / 0.18215)
        t_dec = time.perf_counter() - t0
    t_step = min(steps)
    total = t_enc + t_dec + 2 * S * t_step
    return {"value": round(1.0 / total, 6), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle (CPU fp32 torch, {cores} threads), ONE {size}x{size} image through the full-size stack: VAE encode {t_enc:.2f} s, "
                      f"one SD-UNet + scheduler step {t_step:.2f} s (best of 2), VAE decode {t_dec:.2f} s; extrapolated to {S}+{S} steps = {total:.0f} s"}


def cpu_baseline_sd_train(P, size):
    """One optimisation step of the oracle SD-2.1 UNet (+ CustomEmbedding) under torch.autograd + clip + AdamW at batch 1."""
    from oracle import CustomEmbeddingRef, SD21_UNET_CONFIG, UNet2DConditionRef
    from oracle import class_emb_to_encoder_hidden_states as ehs_ref
    cores = _cpu_threads()
    torch.manual_seed(0)
    unet, emb = UNet2DConditionRef(**SD21_UNET_CONFIG), CustomEmbeddingRef(2, 1024)
    params = list(unet.parameters()) + list(emb.parameters())
    opt = torch.optim.AdamW(params, lr=1e-5, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    g = torch.Generator().manual_seed(3)
    lat, target = torch.randn(1, 4, size, size, generator=g), torch.randn(1, 4, size, size, generator=g)
    labels, ts = torch.zeros(1, dtype=torch.long), torch.tensor([500])

    def step():
        out = unet(lat, ts, ehs_ref(emb(labels))).sample
        loss = torch.nn.functional.mse_loss(out, target)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
    t0 = time.perf_counter()
    step()
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    step()
    dt = min(first, time.perf_counter() - t0)
    return {"value": round(1.0 / dt, 5), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"2 optimisation steps (best taken) of the CPU oracle SD-2.1 UNet + CustomEmbedding (torch.autograd + clip_grad_norm_ + "
                      f"AdamW) at batch 1, {size}x{size} latents, fp32, {cores} threads: {dt:.1f} s per step"}


def main_train(args, P, world, rank, dev, dist):
    """configs[1]: DDIM training of cond_unet_2d at 128x128, bf16, data-parallel.  One step = sampling (noise, timesteps,
    add_noise) + forward + loss + backward + bucketed gradient all-reduce (N > 1) + clip/AdamW/EMA + weight re-pack."""
    from phendiff_amd.training import scaled_lr
    B, size = args.batch or 112, args.size or 128
    torch.manual_seed(0)
    unet = P.CustomCondUNet2DModel(compute_dtype=args.dtype, **dict(P.UNET_CONFIGS[args.model], sample_size=size))
    state_dict = {k: v.clone() for k, v in unet.state_dict().items()}
    sched = P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"])
    tr = P.UNetTrainer(unet.to(dev), sched, lr=scaled_lr(1e-4, world))
    clean, labels = synth_batch(B, size, 1234 + rank)
    clean, labels = clean.to(dev), labels.to(dev)
    gen = torch.Generator(device=dev).manual_seed(7 + rank)

    def step():
        noise = torch.randn(clean.shape, device=dev, generator=gen)
        ts = torch.randint(0, sched.config.num_train_timesteps, (B,), device=dev, generator=gen)
        noisy = sched.add_noise(clean, noise, ts)
        return tr.step(noisy, ts, clean, noise, class_labels=labels)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed, ranks = reduce_elapsed(dist, elapsed, dev, B * args.steps)
    if dist is None and not args.no_box:      # clock / power under this workload, through a repeat of the steps AFTER the timed region
        def _again():
            for _ in range(min(args.steps, 3)):
                step()
            torch.cuda.synchronize(dev)
        ranks = dict(ranks, box=SmiSampler.bracket(_again))
    assert torch.isfinite(loss).all()
    value = world * B * args.steps / elapsed
    res = {
        "metric": f"DDIM training images/sec (128x128 cond_unet_2d, {args.dtype})", "value": round(value, 3), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic", **ranks,
        "config": {"workload": f"configs[1]: {size}x{size} cond_unet_2d DDIM training, {args.model} UNet (random init, seed 0), "
                               f"3k_steps_clipping_rescaling / v_prediction, batch {B}/GPU (launch_script_DDIM.sh:52) on {world} GPU(s), "
                               "AdamW(.95,.999) + clip 1.0 + EMA, data-parallel gradient all-reduce overlapped with the backward",
                   "batch_per_gpu": B, "global_batch": B * world, "image_size": size, "final_loss": round(float(loss), 5)},
    }
    if rank == 0 and not args.no_roofline:
        plan = tr.plan_for(B, size, size)
        st = torch.cuda.current_stream(dev).cuda_stream
        prof = {}
        for title, ops in (("fwd", plan.ops), ("bwd", plan.bwd_ops)):
            for k, d in plan._profile_ops(ops, st, reps=2).items():
                prof[f"{title}.{k}"] = d
        torch.cuda.synchronize(dev)
        total_ms = sum(d["ms"] for d in prof.values())
        kind, d = max(prof.items(), key=lambda kv: kv[1]["ms"])
        mfma = any(t in kind for t in ("conv", "wgrad", "dgrad", "attn"))
        if mfma:
            ach, peak, unit = d["flops"] / (d["ms"] * 1e-3) / 1e12, PEAK_MFMA_TFLOPS[args.dtype], "TFLOP/s"
        else:
            ach, peak, unit = d["bytes"] / (d["ms"] * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
        groups, tnote = _pmc_traffic(prof, "train") if (B, size, args.model) == (112, 128, "super_small") else ({}, "PMC passes exist for the default shape only")
        traffic = next((e["hbm_bytes_per_launch"] for e in groups.values() if kind in e["kinds"]), None)
        textra = {"traffic_by_group": {g: {k: e[k] for k in ("kinds", "hbm_bytes_per_launch", "ratio")} for g, e in groups.items()}} if groups else {}
        if tnote:
            textra["traffic_note"] = tnote
        res["roofline"] = {"kernel": kind, "bound": "mfma" if mfma else "hbm", "achieved": round(ach, 2), "peak": peak, "unit": unit,
                           "frac": round(ach / peak, 4), "traffic": traffic, **textra,
                           "launches_per_step": round(d["launches"]), "avg_launch_ms": round(d["ms"] / max(d["launches"], 1), 4),
                           "share_of_fwd_bwd": round(d["ms"] / total_ms, 3),
                           "method": "HIP events between consecutive launches of one forward + backward (same plan and buffers)",
                           "per_kernel_ms": {k: round(v["ms"], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
                           "per_kernel_tflops": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) for k, v in prof.items()
                                                 if v["ms"] > 0 and v["flops"] > 0}}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_train(args.model, size, state_dict, args.cpu_baseline_seconds)
        res["gpu_over_cpu"] = round(value / res["cpu_baseline"]["value"], 1)
    if dist is not None:
        res["data_parallel"] = comm_step_stats(args, dev, dist, tr, step)
    if rank == 0:
        res["diagnostic_env"] = diagnostic_env()
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _sd_stack(P, args, dev, latent_only=False):
    """Full-size latent-diffusion stack, random init (seed 0): SD-2.1 UNet (865.9 M), SD VAE (83.7 M), CustomEmbedding(2, 1024)."""
    torch.manual_seed(0)
    with torch.device(dev):
        unet = P.SDUNet2DConditionModel(compute_dtype=args.dtype, **P.SD21_UNET_CONFIG)
        emb = P.CustomEmbedding(2, 1024)
        vae = None if latent_only else P.AutoencoderKL(compute_dtype=args.dtype)
    return unet, vae, emb, P.DDIMScheduler(**P.SCHEDULER_CONFIGS["SD_orig_config"])


_SELFTEST = {}      # comm_selftest()'s result (world > 1), attached to the line by reduce_elapsed


def reduce_elapsed(dist, elapsed, dev, units_per_rank):
    """MAX of the ranks' elapsed time (the contract's clock) + what each rank did: per-rank units/s and the size of the process
    group RCCL actually formed."""
    if dist is None:
        return elapsed, {"rccl_world_size": 1, "per_rank_units_per_s": [round(units_per_rank / elapsed, 4)]}
    world = dist.get_world_size()
    mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    times = [float(t.item()) for t in every]
    st = _SELFTEST.get("result") or {}
    return max(times), {"rccl_world_size": st.get("rccl_world_size", world), "rccl_world_size_source": st.get("rccl_world_size_source", "torch.distributed.get_world_size()"),
                        "per_rank_units_per_s": [round(units_per_rank / t, 4) for t in times],
                        **({"allreduce_selftest": st} if st else {})}


class _StubComm:
    """Test double of phendiff_amd.comm.NativeComm for the self-test's control flow (tests/test_distributed_gloo.py,
    tests/test_gpu_bench_two_ranks.py): PD_BENCH_SELFTEST_STUB="<rank>:<seconds>" makes pd_comm_init's stand-in sleep that long on that
    rank ("-1:0": nobody sleeps); its all-reduce is never reached when a rank timed out, and sums through the torch group otherwise."""
    spec = (-1, 0.0)

    def __init__(self, rank, world, comm_id, device=None):
        assert len(comm_id) == 128
        self.rank, self.world = rank, world
        if rank == self.spec[0]:
            time.sleep(self.spec[1])

    @staticmethod
    def unique_id():
        return bytes(range(128))

    def query(self):
        return self.rank, self.world

    def allreduce_(self, flat, mean=True, algo=1, stream=None):
        import torch.distributed as dist
        dist.all_reduce(flat)
        return flat

    def close(self):
        pass


def comm_selftest(dist, dev, nbytes=64 << 20, native_timeout_s=90.0, native=None, comm_cls=None):
    """world > 1, before anything is timed: one 64 MB fp32 bucket all-reduced through ``torch.distributed.all_reduce`` (RCCL through
    PyTorch), checked bit for bit against the analytic sum and timed; its bus bandwidth 2 (W - 1) / W x bytes / t goes on the JSON line.
    If the scaling curve bends, this says what the exchange can do on that node.

    The two legs through the C ABI -- ``pd_allreduce_bucket`` algo 0 (ncclAllReduce) and algo 1 (reduce-scatter + all-gather), with the
    size of the communicator as RCCL itself reports it (``pd_comm_query``: ncclCommCount) -- are OPT-IN (``PD_BENCH_NATIVE_SELFTEST=1`` or
    ``native=True``): the driver's default command runs the torch leg only, because the native communicator has never been formed
    with more than one rank anywhere and the self-test must not be able to cost the run it explains.  When they
    run, every torch collective is issued by the MAIN thread and every rank issues the same sequence of them whatever fails where:
      * rank 0 draws the 128-byte id (failure -> a sentinel) and ONE ``dist.broadcast`` of 129 bytes carries it: every rank learns
        together whether there is an id (no rank is left alone inside a broadcast);
      * only ``pd_comm_init`` (ncclCommInitRank, which can stall) runs in a worker thread, with a time limit; a thread that does not
        come back is LEFT (daemon; it touches neither torch nor the process group) and its communicator is abandoned, not destroyed;
      * one ``all_reduce(MIN)`` of a status code makes the ranks agree on ok / error / timeout; each native leg then runs inside
        try / except with one more agreed status, so an error on one rank ends the native part on all of them."""
    import threading
    world, rank = dist.get_world_size(), dist.get_rank()
    cuda = dev.type == "cuda"
    n = nbytes // 4
    n -= n % (world * 256)
    base = (torch.arange(n, device=dev, dtype=torch.float32) % 1024) / 1024          # exact dyadic fractions
    want = base * float(world * (world + 1) // 2)                                   # sum over ranks of (rank + 1) * base: exact in fp32
    out = {"bytes": n * 4, "world": world, "busbw_GBs": {}, "exact": {}}
    rehearsal = bool(os.environ.get("PD_BENCH_REHEARSAL"))

    def sync():
        if cuda:
            torch.cuda.synchronize(dev)

    def agree(code):
        """MIN over the ranks of a small status code: one torch collective, main thread, every rank."""
        flag = torch.tensor([code], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item())

    def timed(fn, reps=3):
        """(exact, busbw GB/s) of one way of all-reducing the bucket, or (error string, None) -- agreed by all ranks."""
        err, ok, ms = None, False, 0.0
        try:
            buf = base * float(rank + 1)
            fn(buf)                                                                 # warm-up (and the checked result)
            sync()
            ok = bool(torch.equal(buf, want))
        except Exception as e:                                                      # noqa: BLE001 -- recorded, agreed below
            err = repr(e)
        if not agree(0 if err else 1):
            return err or "failed on another rank", None
        dist.barrier()
        try:
            bufs = [base * float(rank + 1) for _ in range(reps)]
            sync()
            t0 = time.perf_counter()
            for b in bufs:
                fn(b)
            sync()
            ms = 1e3 * (time.perf_counter() - t0) / reps
        except Exception as e:                                                      # noqa: BLE001
            err = repr(e)
        if not agree(0 if err else 1):
            return err or "failed on another rank", None
        return ok, round(2.0 * (world - 1) / world * n * 4 / (ms * 1e-3) / 1e9, 2)

    out["exact"]["torch"], out["busbw_GBs"]["torch"] = timed(lambda b: dist.all_reduce(b, op=dist.ReduceOp.SUM))
    out["rccl_world_size"], out["rccl_world_size_source"] = world, "torch.distributed.get_world_size()"
    stub = os.environ.get("PD_BENCH_SELFTEST_STUB")
    if comm_cls is None and stub:
        r_, s_ = stub.split(":")
        _StubComm.spec = (int(r_), float(s_))
        comm_cls, native = _StubComm, True
    if native is None:
        native = os.environ.get("PD_BENCH_NATIVE_SELFTEST") == "1"
    if not native:
        out["native"] = "not run: the C-ABI legs are opt-in (PD_BENCH_NATIVE_SELFTEST=1)"
        return out
    if rehearsal and comm_cls is None:
        out["native"] = "skipped: rehearsal over gloo on one device (RCCL refuses two ranks per device)"
        return out
    if comm_cls is None:
        from phendiff_amd.comm import NativeComm as comm_cls
    # the id: drawn on rank 0, shipped by the main thread; byte 0 says whether there is one
    msg = torch.zeros(129, dtype=torch.uint8, device=dev)
    id_err = None
    if rank == 0:
        try:
            cid = comm_cls.unique_id()
            msg[0] = 1
            msg[1:] = torch.frombuffer(bytearray(cid), dtype=torch.uint8).to(dev)
        except Exception as e:                                                      # noqa: BLE001 -- the sentinel goes out instead
            id_err = repr(e)
    dist.broadcast(msg, src=0)
    host = msg.cpu()
    if int(host[0]) == 0:
        out["native"] = "no communicator id: pd_comm_unique_id failed on rank 0" + (f" ({id_err})" if id_err else "")
        return out
    cid = bytes(host[1:].tolist())
    box = {}

    def init():                                                                     # pd_comm_init only: no torch collective in here
        try:
            box["comm"] = comm_cls(rank, world, cid, dev if cuda else None)
        except Exception as e:                                                      # noqa: BLE001 -- recorded, never fatal
            box["error"] = repr(e)

    native_timeout_s = float(os.environ.get("PD_BENCH_NATIVE_TIMEOUT_S", native_timeout_s))
    th = threading.Thread(target=init, daemon=True)
    th.start()
    th.join(native_timeout_s)
    status = agree(2 if "comm" in box else (1 if "error" in box else 0))            # 2 ok / 1 error / 0 still inside pd_comm_init
    if status < 2:
        out["native"] = "timeout" if status == 0 else "error"
        out["native_detail"] = (f"pd_comm_init did not return within {native_timeout_s:.0f} s on some rank; the thread is left behind, its "
                                "communicator abandoned") if status == 0 else box.get("error", "pd_comm_init failed on another rank")
        comm = box.pop("comm", None)
        if comm is not None:
            comm._comm = None            # a communicator whose peers are missing is not destroyed (ncclCommDestroy may wait for them)
        return out
    comm = box["comm"]
    try:
        r_, w_ = comm.query()
        out["rccl_world_size"], out["rccl_world_size_source"] = w_, "pd_comm_query (ncclCommCount)"
        out["rccl_rank_matches"] = bool(r_ == rank)
    except Exception as e:                                                          # noqa: BLE001
        out["native_detail"] = repr(e)
    for name, algo in (("rccl_allreduce", 0), ("rs_ag", 1)):
        out["exact"][name], out["busbw_GBs"][name] = timed(lambda b, algo=algo: comm.allreduce_(b, mean=False, algo=algo))
        if out["busbw_GBs"][name] is None:
            break                                                                   # an agreed failure ends the native part everywhere
    out["native"] = "ran"
    try:
        comm.close()
    except Exception:                                                               # noqa: BLE001
        pass
    return out


def comm_step_stats(args, dev, dist, tr, step, steps=3):
    """world > 1, after the timed region of a training workload: (i) `allreduce_overlap_frac` -- the same step with events around every
    bucket's collective on the comm stream: comm-stream busy time hidden under the backward / comm time; (ii) `step_ms_no_comm` --
    the same step with the exchange left out (ranks then drift apart: these steps run last and their weights are discarded)."""
    out = {}

    def run(k):
        torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        torch.cuda.synchronize(dev)
        return 1e3 * (time.perf_counter() - t0) / k
    tr.comm_timing = True
    step()
    torch.cuda.synchronize(dev)
    stats = []
    for _ in range(steps):
        step()
        torch.cuda.synchronize(dev)
        stats.append(tr.comm_overlap_stats())
    tr.comm_timing = False
    stats = [s for s in stats if s]
    if stats:
        mid = sorted(stats, key=lambda s: s["overlap_frac"] or 0.0)[len(stats) // 2]
        out["allreduce_overlap_frac"] = mid["overlap_frac"]
        out["allreduce"] = mid
    out["step_ms_with_comm"] = round(run(steps), 3)
    tr.skip_collectives = True
    step()
    out["step_ms_no_comm"] = round(run(steps), 3)
    tr.skip_collectives = False
    mine = torch.tensor([out.get("allreduce_overlap_frac") or 0.0, out["step_ms_with_comm"], out["step_ms_no_comm"]], dtype=torch.float64, device=dev)
    every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(every, mine)
    out["per_rank"] = {"allreduce_overlap_frac": [round(float(t[0]), 4) for t in every], "step_ms_with_comm": [round(float(t[1]), 3) for t in every],
                       "step_ms_no_comm": [round(float(t[2]), 3) for t in every]}
    return out


def _timed_steps(args, dev, dist, step, units_per_rank):
    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed, ranks = reduce_elapsed(dist, elapsed, dev, units_per_rank)
    if dist is None and not getattr(args, "no_box", False):
        # the clock / power the card holds under THIS workload (round 6: the latent-diffusion legs run at the board's power cap too --
        # LAB_r6 section 1), sampled through a repeat of up to 3 steps after the timed region, never inside it
        def _again():
            for _ in range(min(args.steps, 3)):
                step()
            torch.cuda.synchronize(dev)
        ranks = dict(ranks, box=SmiSampler.bracket(_again))
    return elapsed, out, ranks


def _back_to_back_ms(ops, stream, reps=5):
    """Total device time of one pass over ``ops`` (launched back to back, no fences in between), averaged over ``reps``."""
    import ctypes as C
    from phendiff_amd import _lib as L
    lib = L.lib()
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.check(lib.pd_event_create(C.byref(e0)), "pd_event_create")
    L.check(lib.pd_event_create(C.byref(e1)), "pd_event_create")
    for op in ops:                                             # warm
        L.check(op.fn(C.byref(op.args), stream), op.what)
    L.check(lib.pd_event_record(e0, stream), "pd_event_record")
    for _ in range(reps):
        for op in ops:
            L.check(op.fn(C.byref(op.args), stream), op.what)
    L.check(lib.pd_event_record(e1, stream), "pd_event_record")
    ms = C.c_float()
    L.check(lib.pd_event_elapsed_ms(e0, e1, C.byref(ms)), "pd_event_elapsed_ms")
    lib.pd_event_destroy(e0)
    lib.pd_event_destroy(e1)
    return ms.value / reps


def _in_situ_launch_ms(plan, kind, stream, forwards=72, discard=12):
    """Average device time of ONE launch of kernel kind ``kind`` INSIDE the steady-state forward: the plan's launches
    are replayed eagerly, forward after forward, exactly as the captured graph holds them (same operands, same buffers; at this
    batch the host runs far ahead of the device, so the stream never drains), with ONE event pair per forward around one launch
    of that kind, rotating over its launches.  The chip therefore sits at the clock / power state the whole instruction mix gives
    it -- not at the boost clock a kernel sees when its launches run back to back in isolation (attention: 1.56 vs 1.72 ms in
    round 4) -- and one fence pair per ~20 ms forward perturbs nothing.  Called right after the timed region (chip hot); the first
    ``discard`` forwards are not counted.  Returns (mean ms, per-launch-slot means, n)."""
    import ctypes as C
    from phendiff_amd import _lib as L
    lib = L.lib()
    ops = plan.ops
    slots = [i for i, op in enumerate(ops) if op.what == kind]
    evs = []
    for _ in range(2 * forwards):
        e = C.c_void_p()
        L.check(lib.pd_event_create(C.byref(e)), "pd_event_create")
        evs.append(e)
    for j in range(forwards):
        k = slots[j % len(slots)]
        for i, op in enumerate(ops):
            if i == k:
                L.check(lib.pd_event_record(evs[2 * j], stream), "pd_event_record")
            L.check(op.fn(C.byref(op.args), stream), op.what)
            if i == k:
                L.check(lib.pd_event_record(evs[2 * j + 1], stream), "pd_event_record")
    by_slot = {}
    ms = C.c_float()
    for j in range(forwards):
        L.check(lib.pd_event_elapsed_ms(evs[2 * j], evs[2 * j + 1], C.byref(ms)), "pd_event_elapsed_ms")      # (synchronises on the event)
        if j >= discard:
            by_slot.setdefault(j % len(slots), []).append(ms.value)
    for e in evs:
        lib.pd_event_destroy(e)
    slot_means = [sum(v) / len(v) for _, v in sorted(by_slot.items())]
    n = sum(len(v) for v in by_slot.values())
    return sum(slot_means) / len(slot_means), slot_means, n


def _pmc_traffic(prof, workload):
    """HBM bytes per launch-plan op from the committed one-step PMC passes of this workload (profiles/r6_hbm_traffic_<workload>.json,
    scripts/collect_traffic_step.py), for every kernel group of scripts/kernel_kinds.py; {} when the file is missing or was measured on
    other kernel sources.  A group covers one or more plan kinds (conv_kernel serves forward convolutions AND input gradients): its
    measured bytes are set against the summed algorithmic bytes and launches of those kinds."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        from kernel_kinds import GROUP_KINDS
        from phendiff_amd._lib import source_hash
        fname = f"r6_hbm_traffic_{workload}.json"
        j = json.load(open(os.path.join(ROOT, "profiles", fname)))
    except (OSError, ValueError, ImportError):
        return {}, f"profiles/r6_hbm_traffic_{workload}.json missing"
    if j.get("sources_sha256") != source_hash():
        return {}, f"profiles/{fname} was measured on other kernel sources (re-run scripts/collect_profiles.sh pmc_side)"
    out = {}
    for grp, e in j.get("groups", {}).items():
        kinds = [k for k in GROUP_KINDS.get(grp, []) if k in prof]
        if not kinds:
            continue
        launches = sum(prof[k]["launches"] for k in kinds)
        alg = sum(prof[k]["bytes"] for k in kinds)
        out[grp] = {"kinds": kinds, "hbm_bytes_per_step": round(e["hbm_bytes"]), "hbm_bytes_per_launch": round(e["hbm_bytes"] / max(launches, 1)),
                    "algorithmic_bytes_per_step": round(alg), "ratio": round(e["hbm_bytes"] / alg, 3) if alg > 0 else None, "source": f"profiles/{fname}"}
    return out, None


def _plan_roofline(prof, dtype, method, workload=None):
    total_ms = sum(d["ms"] for d in prof.values())
    kind, d = max(prof.items(), key=lambda kv: kv[1]["ms"])
    mfma = d["flops"] > 0
    if mfma:
        ach, peak, unit = d["flops"] / (d["ms"] * 1e-3) / 1e12, PEAK_MFMA_TFLOPS[dtype], "TFLOP/s"
    else:
        ach, peak, unit = d["bytes"] / (d["ms"] * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
    traffic, tnote, groups = None, None, {}
    if workload is not None:
        groups, tnote = _pmc_traffic(prof, workload)
        for grp, e in groups.items():
            if kind in e["kinds"]:
                traffic = e["hbm_bytes_per_launch"]
                if len(e["kinds"]) > 1:
                    tnote = f"group {grp} = kinds {e['kinds']} (one kernel template serves them all): bytes per launch averaged over the group"
    extra = {"traffic_by_group": {g: {k: e[k] for k in ("kinds", "hbm_bytes_per_launch", "ratio")} for g, e in groups.items()}} if groups else {}
    if tnote:
        extra["traffic_note"] = tnote
    return {"kernel": kind, "bound": "mfma" if mfma else "hbm", "achieved": round(ach, 2), "peak": peak, "unit": unit,
            "frac": round(ach / peak, 4), "traffic": traffic, **extra, "launches": round(d["launches"]),
            "avg_launch_ms": round(d["ms"] / max(d["launches"], 1), 4), "share": round(d["ms"] / total_ms, 3), "method": method,
            "per_kernel_ms": {k: round(v["ms"], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
            "per_kernel_tflops": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) for k, v in prof.items() if v["ms"] > 0 and v["flops"] > 0}}


def main_sd_img2img(args, P, world, rank, dev, dist):
    """configs[4]: custom_pipeline_stable_diffusion_img2img DDIB at 512x512 (64x64 latents): VAE encode -> S-step DDIM inversion
    under the original class -> class swap -> S-step denoising -> VAE decode.  Images are independent: sharded, no collective."""
    B, size, S = args.batch or 32, args.size or 512, args.inference_steps     # B = 8 / 16 / 32 / 64: 4.4 / 6.1 / 7.2 / 7.4 images/s
    unet, vae, emb, sched = _sd_stack(P, args, dev)
    pipe = P.CustomStableDiffusionImg2ImgPipeline(vae, unet, sched, emb)
    x, labels = synth_batch(B, size, 1234 + rank)
    x, labels = x.to(dev), labels.to(dev)
    gen = torch.Generator(device=dev).manual_seed(7 + rank)
    host_out = torch.empty((B, size, size, 3), dtype=torch.float32, pin_memory=True)
    if args.no_graph:
        def step():
            host_out.copy_(torch.from_numpy(P.ddib(pipe, x, labels, 1 - labels, S, generator=gen)))
            return host_out
        P.ddib(pipe, x, labels, 1 - labels, 1, generator=gen)          # builds every launch plan
    else:
        # the whole transfer (VAE encode -> 2*S SD-UNet steps -> VAE decode, ~36 000 launches) as ONE hipGraph, bit-identical to
        # the eager loop (tests/test_gpu_sd_pipeline.py); output lands in pinned host memory inside the timed region
        graph = P.SDDDIBGraph(pipe, batch_size=B, num_inference_steps=S, height=size, width=size)

        def step():
            host_out.copy_(graph.run(x, labels, 1 - labels, generator=gen).images, non_blocking=True)
            return host_out
    elapsed, out, ranks = _timed_steps(args, dev, dist, step, B * args.steps)
    assert out.shape == (B, size, size, 3) and torch.isfinite(out).all()
    value = world * B * args.steps / elapsed
    res = {"metric": "SD img2img images/sec (VAE encode + 50-step DDIM invert + 50-step denoise + VAE decode, 512x512)",
           "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1000 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": args.dtype, "data": "synthetic", **ranks,
           "config": {"workload": f"configs[4]: custom_pipeline_stable_diffusion_img2img DDIB, {size}x{size} images ({size // 8}x{size // 8} "
                                  f"latents), {S}+{S} DDIM steps, SD-2.1 UNet (865.9 M) + SD VAE (83.7 M) + CustomEmbedding, random init, "
                                  f"SD_orig_config / v_prediction, {B} images/GPU/step sharded over {world} GPU(s), no collectives",
                      "batch_per_gpu": B, "global_batch": B * world, "inference_steps": S, "image_size": size,
                      "hipgraph": not args.no_graph}}
    if rank == 0 and not args.no_roofline:
        plan = next(p for k, p in unet._plans.items() if k[0] == B)
        prof = plan._profile_ops(plan.ops, torch.cuda.current_stream(dev).cuda_stream, reps=2)
        res["roofline"] = _plan_roofline(prof, args.dtype, "HIP events between consecutive launches of one SD-UNet forward (same plan "
                                                           "and buffers as the timed region; 2*S of them per step)", workload="sd_img2img")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_sd_img2img(P, size, S)
        res["gpu_over_cpu"] = round(value / res["cpu_baseline"]["value"], 1)
    if rank == 0:
        res["diagnostic_env"] = diagnostic_env()
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main_sd_train(args, P, world, rank, dev, dist):
    """configs[3]: SD-2.1 UNet + CustomEmbedding fine-tuning at 64x64 latents (512 px), data-parallel.  One step = sampling
    (noise, timesteps, add_noise on the latents) + _SD_prediction_wrapper forward + loss + backward + bucketed gradient
    all-reduce (3.46 GB fp32, N > 1) + clip/AdamW/EMA + weight re-pack; every 10th step is unconditional (proba_uncond = 0.1)."""
    from phendiff_amd.training import scaled_lr
    B, size = args.batch or 32, args.size or 64        # B = 8 / 16 / 32 / 64: 106 / 153 / 196 / 216 samples/s (288 GB of HBM: use it)
    unet, _, emb, sched = _sd_stack(P, args, dev, latent_only=True)
    tr = P.SDUNetTrainer(unet, emb, sched, lr=scaled_lr(1e-5, world))
    g = torch.Generator().manual_seed(1234 + rank)
    clean = (torch.randn(B, 4, size, size, generator=g) * 0.8).to(dev)       # VAE latents * scaling_factor have ~unit scale
    labels = (torch.arange(B) % 2).to(dev)
    gen = torch.Generator(device=dev).manual_seed(7 + rank)
    count = [0]

    def step():
        noise = torch.randn(clean.shape, device=dev, generator=gen)
        ts = torch.randint(0, sched.config.num_train_timesteps, (B,), device=dev, generator=gen)
        noisy = sched.add_noise(clean, noise, ts)
        count[0] += 1
        return tr.step(noisy, ts, clean, noise, labels, unconditional=(count[0] % 10 == 0))

    elapsed, loss, ranks = _timed_steps(args, dev, dist, step, B * args.steps)
    assert torch.isfinite(loss).all()
    value = world * B * args.steps / elapsed
    res = {"metric": f"SD-2.1 UNet fine-tuning samples/sec (64x64 latents = 512x512, {args.dtype})", "value": round(value, 3), "unit": "samples/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / args.steps, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic", **ranks,
           "config": {"workload": f"configs[3]: SD-2.1 UNet (865.9 M, random init) + CustomEmbedding fine-tune, {size}x{size} latents, "
                                  f"batch {B}/GPU on {world} GPU(s), SD_orig_config / v_prediction, AdamW(.95,.999) + clip 1.0 + EMA, "
                                  "data-parallel gradient all-reduce (64 MB buckets) overlapped with the backward",
                      "batch_per_gpu": B, "global_batch": B * world, "latent_size": size, "final_loss": round(float(loss), 5)}}
    if rank == 0 and not args.no_roofline:
        plan = tr.plan_for(B, size, size)
        st = torch.cuda.current_stream(dev).cuda_stream
        prof = {}
        for title, ops in (("fwd", plan.ops), ("bwd", plan.bwd_ops)):
            for k, d in plan._profile_ops(ops, st, reps=1).items():
                prof[f"{title}.{k}"] = d
        torch.cuda.synchronize(dev)
        res["roofline"] = _plan_roofline(prof, args.dtype, "HIP events between consecutive launches of one forward + backward (same plan and buffers)", workload=args.workload)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_sd_train(P, size)
        res["gpu_over_cpu"] = round(value / res["cpu_baseline"]["value"], 1)
    if dist is not None:
        res["data_parallel"] = comm_step_stats(args, dev, dist, tr, step)
    if rank == 0:
        res["diagnostic_env"] = diagnostic_env()
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# Kernel-selecting diagnostic overrides the library reads from the environment (conv_igemm.hip / linear_gemm.hip / attn_d8.hip /
# unet.py): a bench line measured under one of them says so, and the default (driver) run is expected to carry none.
DIAG_ENV = ("PD_LIB", "PD_TW_DMA", "PD_ALLOW_ABI_MISMATCH", "PD_BENCH_REHEARSAL", "PD_LIN_DMA", "PD_CONV_NCO", "PD_CONV_PLAIN", "PD_LIN_NC4", "PD_ATTN_NO_GLDS", "PD_ATTN_LDS_PAD", "PD_PREAPPLY_MIN_COUT", "PD_NO_LINEAR_GRADS", "PD_NO_PREAPPLY_WGRAD",
            "PD_CONV_PRO", "PD_SUBPIXEL_UP", "PD_LIN_FOLD", "PD_LIN_NC5", "PD_LIN_P8", "PD_TW_XCD", "PD_CONV_XCD", "PD_ATTN64_BWD_XCD", "PD_BENCH_NO_SELFTEST", "PD_ATTN_BWD_FUSED", "PD_GN_FUSED", "PD_BENCH_NATIVE_SELFTEST", "PD_BENCH_SELFTEST_STUB", "PD_BENCH_NATIVE_TIMEOUT_S", "PD_ATTN_WPB4", "PD_ATTN64_QB1", "EXTRA_HIPCC_FLAGS")


def diagnostic_env():
    return {k: os.environ[k] for k in DIAG_ENV if k in os.environ}


# BASELINE.json configs[1], [3], [4] as short legs behind the headline: one child process each (fresh
# allocator, a crash or a timeout cannot take the headline line down), rank 0 / world 1 only, bounded by --side-budget-s.
SIDE_WORKLOADS = (
    # name,            argv,                                                                             expected seconds
    ("train",           ["--workload", "train", "--steps", "5", "--warmup", "2", "--cpu-baseline-seconds", "10"], 60),
    ("sd_train",        ["--workload", "sd_train", "--steps", "3", "--warmup", "1"], 120),
    ("sd_img2img_fp16", ["--workload", "sd_img2img", "--dtype", "fp16", "--steps", "1", "--warmup", "1"], 150),
    ("sd_img2img_bf16", ["--workload", "sd_img2img", "--dtype", "bf16", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], 90),
    # round 6: configs[2] and configs[1] at the reference's own precision, `mixed_precision: fp16`
    # (examples/example_img2img_comparison_conf/general_config.yaml:46, examples/examples_training_scripts/launch_script_DDIM.sh:56):
    # same batch, same graph replay / LossScaler step, own in-situ roofline; the CPU baseline is the fp32 oracle either way (shared)
    ("img2img_fp16",    ["--workload", "img2img", "--dtype", "fp16", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-sweep"], 60),
    ("train_fp16",      ["--workload", "train", "--dtype", "fp16", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"], 45),
)


def run_side_workloads(budget_s, only=None, headline_cpu_baseline=None):
    """Each leg = `python bench.py --workload X ... --no-side-workloads` as a child; its JSON line is attached (trimmed of the
    long prose fields).  A leg that does not fit what is left of the budget is skipped and the line says so."""
    import subprocess
    out, t_all = {}, time.time()
    for name, argv, expect_s in SIDE_WORKLOADS:
        if only and name not in only:
            continue
        left = budget_s - (time.time() - t_all)
        if left < expect_s * 0.6:
            out[name] = {"skipped": f"side budget: {left:.0f} s left of {budget_s:.0f}, this leg needs ~{expect_s} s"}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-side-workloads"] + argv
        t0 = time.time()
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=max(left, expect_s) + 60)
        except subprocess.TimeoutExpired:
            out[name] = {"failed": f"timeout after {time.time() - t0:.0f} s"}
            continue
        line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
        if r.returncode != 0 or line is None:
            out[name] = {"failed": f"rc {r.returncode}", "stderr_tail": r.stderr[-400:]}
            continue
        j = json.loads(line)
        roof = j.get("roofline") or {}
        keep_roof = {k: roof[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_by_group", "traffic_note", "launches", "launches_per_step",
                                           "avg_launch_ms", "share", "share_of_fwd_bwd", "per_kernel_ms", "per_kernel_tflops") if k in roof}
        out[name] = {"metric": j["metric"], "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"],
                     "warmup": j["warmup"], "dtype": j["dtype"], "n_gpus": j["n_gpus"], "workload": j["config"]["workload"],
                     "batch_per_gpu": j["config"].get("batch_per_gpu"), "roofline": keep_roof,
                     "cpu_baseline": j.get("cpu_baseline"), "gpu_over_cpu": j.get("gpu_over_cpu"), "box": j.get("box"),
                     "leg_wall_s": round(time.time() - t0, 1)}
    # the latent-diffusion CPU baseline is the fp32 oracle whatever the engine's storage type: timed once (fp16 leg), quoted on both
    a, b = out.get("sd_img2img_fp16", {}), out.get("sd_img2img_bf16", {})
    if a.get("cpu_baseline") and "value" in b and not b.get("cpu_baseline"):
        b["cpu_baseline"] = dict(a["cpu_baseline"], shared="timed once in the sd_img2img_fp16 leg (the oracle is fp32 either way)")
        b["gpu_over_cpu"] = round(b["value"] / a["cpu_baseline"]["value"], 1)
    for leg, src, where in (("train_fp16", out.get("train", {}).get("cpu_baseline"), "the train leg"),
                            ("img2img_fp16", headline_cpu_baseline, "the headline")):
        b = out.get(leg, {})
        if src and "value" in b and not b.get("cpu_baseline"):
            b["cpu_baseline"] = dict(src, shared=f"timed once in {where} (the oracle is fp32 either way)")
            b["gpu_over_cpu"] = round(b["value"] / src["value"], 1)
    out["total_wall_s"] = round(time.time() - t_all, 1)
    return out


def launcher_command(gpus, argv, port):
    """The command line `python bench.py --gpus N ...` turns into: one process per GPU under torch.distributed.run, rendezvous on
    127.0.0.1 (the container hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(gpus, argv):
    """Start `gpus` fresh ranks of this script and return their exit code.  Called before anything initialises the GPU in this
    process (counting devices does not), and the ranks are child processes, never an exec of a GPU-touched one."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = launcher_command(gpus, argv, port)
    if os.environ.get("PD_BENCH_PRINT_LAUNCH"):         # tests: show the command line instead of running it
        print(json.dumps(cmd))
        return 0
    have = torch.cuda.device_count()
    if have < gpus:
        print(f"bench.py: --gpus {gpus} but this node exposes {have} GPU(s)", file=sys.stderr)
        return 2
    return subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="img2img", choices=["img2img", "train", "sd_img2img", "sd_train"],
                    help="img2img = BASELINE.json's metric (default); train = configs[1], one optimisation step per 'step'; "
                         "sd_img2img = configs[4] (latent-diffusion DDIB at 512x512); sd_train = configs[3] (SD-2.1 UNet fine-tuning step)")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (default 32; train: 112 = launch_script_DDIM.sh:52)")
    ap.add_argument("--size", type=int, default=None, help="image size (default 256; train: 128)")
    ap.add_argument("--inference-steps", type=int, default=50)
    ap.add_argument("--model", default="super_small")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "f32"],
                    help="engine mode; fp16 = the reference's --mixed_precision fp16 (img2img, sd_img2img, and train under a loss scale)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--streams", type=int, default=1, help="split the per-GPU batch into this many concurrently replayed "
                    "trajectories (separate HIP streams); measured: no gain (DESIGN.md section 6, scripts/bench_concurrent.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", dest="sweep", action="store_false", help="img2img: skip the B = 16 / 64 side measurements")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-box", action="store_true", help="skip the rocm-smi clock / power samples (they run through a REPEAT of the timed steps "
                    "after the timed region: profiled runs that count batches or bytes per step must not see those extra steps)")
    ap.add_argument("--no-side-workloads", dest="side_workloads", action="store_false",
                    help="img2img on 1 GPU: skip the short legs of configs[1] (train), configs[3] (sd_train) and configs[4] (sd_img2img, "
                         "fp16 and bf16) that are attached to the line as `side_workloads`")
    ap.add_argument("--side-budget-s", type=float, default=330.0, help="wall-clock bound for all side workloads together")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=30.0, help="bound of the cpu_baseline sample (img2img, train)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher (the reference's `accelerate launch --multi_gpu --num_processes N`,
        # launch_script_DDIM.sh:19-34).  Nothing in this process has touched the GPU: the ranks are fresh children.
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    import phendiff_amd as P
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree", file=sys.stderr)
        sys.exit(2)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # PD_BENCH_REHEARSAL=1 (tests/test_gpu_bench_two_ranks.py): N ranks on the ONE GPU of a test box over gloo -- RCCL refuses two
        # ranks per device -- so that the whole N > 1 control flow (barriers, the all_gather of the ranks' clocks, rank 0's line)
        # is exercised before the driver's 8-GPU run.  The line says so (`rehearsal`); never a measurement.
        try:
            if os.environ.get("PD_BENCH_REHEARSAL"):
                local_rank = local_rank % max(torch.cuda.device_count(), 1)
                dist.init_process_group("gloo")
            else:
                if os.environ.get("PD_BENCH_FAIL_INIT") == str(rank):          # tests: this rank's process group cannot be formed
                    raise RuntimeError("PD_BENCH_FAIL_INIT: simulated init_process_group failure")
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        except Exception as e:                                   # noqa: BLE001
            # a rank that cannot join the process group has nothing to time: one line, non-zero exit (the launcher then ends the others);
            # nothing here re-executes a process that touched the GPU
            print(f"bench.py: rank {rank}/{world}: init_process_group failed: {type(e).__name__}: {str(e).splitlines()[0] if str(e) else ''}",
                  file=sys.stderr, flush=True)
            sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist is not None and not os.environ.get("PD_BENCH_NO_SELFTEST"):
        try:
            _SELFTEST["result"] = comm_selftest(dist, dev)
        except Exception as e:                                                     # noqa: BLE001 -- the self-test never takes the run down
            _SELFTEST["result"] = {"failed": repr(e)}
    if args.workload == "train":
        return main_train(args, P, world, rank, dev, dist)
    if args.workload == "sd_img2img":
        return main_sd_img2img(args, P, world, rank, dev, dist)
    if args.workload == "sd_train":
        return main_sd_train(args, P, world, rank, dev, dist)
    args.batch, args.size = args.batch or 32, args.size or 256

    B, S, size = args.batch, args.inference_steps, args.size
    torch.manual_seed(0)  # identical random-init weights on every rank (no checkpoint: no network)
    unet = P.CustomCondUNet2DModel(compute_dtype=args.dtype, **dict(P.UNET_CONFIGS[args.model], sample_size=size))
    state_dict = {k: v.clone() for k, v in unet.state_dict().items()}
    pipe = P.ConditionalDDIMPipeline(unet.to(dev), P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"]))
    assert B % args.streams == 0
    Bs = B // args.streams
    # each concurrent trajectory owns its activation buffers (private_plan)
    runners = [P.DDIBGraph(pipe, batch_size=Bs, num_inference_steps=S, use_graph=not args.no_graph, private_plan=i > 0)
               for i in range(args.streams)]
    runner = runners[0]
    x, labels = synth_batch(B, size, 1234 + rank)
    x, labels = x.to(dev), labels.to(dev)
    target = P.swap_binary_labels(labels)

    class _Multi:
        def run(self, x, labels, target):
            for i, r_ in enumerate(runners):
                sl = slice(i * Bs, (i + 1) * Bs)
                r_.run(x[sl], labels[sl], target[sl], join=False)
            for r_ in runners:
                r_.join()
    multi = _Multi()

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # the timed region runs from the device-resident input batch to the HOST output array (the reference's one D2H
    # of B*H*W*3 floats per batch, pipeline_conditionial_ddim.py:349-350): pinned buffer, asynchronous copy behind each batch
    host_out = torch.empty((B, size, size, 3), dtype=torch.float32, pin_memory=True)

    def one_batch():
        multi.run(x, labels, target)
        for i, r_ in enumerate(runners):
            host_out[i * Bs:(i + 1) * Bs].copy_(r_.images, non_blocking=True)

    for _ in range(args.warmup):
        one_batch()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_batch()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed, ranks = reduce_elapsed(dist, elapsed, dev, B * args.steps)
    box = None
    if rank == 0 and world == 1 and not args.no_box:          # (one-GPU runs only: at N > 1 rank 0 would keep the other ranks waiting at the next barrier)
        def _again():
            for _ in range(min(args.steps, 3)):
                one_batch()
            torch.cuda.synchronize(dev)
        box = SmiSampler.bracket(_again)
    assert torch.isfinite(host_out).all() and float(host_out.min()) >= 0.0 and float(host_out.max()) <= 1.0

    images = world * B * args.steps
    value = images / elapsed
    res = {
        "metric": f"img2img images/sec ({S}-step DDIM invert+denoise, {size}x{size})", "value": round(value, 4), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic", **ranks,
        "config": {"workload": f"configs[2]: {size}x{size} pipeline_conditional_ddim invert->class-swap->denoise, "
                               f"{S}+{S} DDIM steps, {args.model} UNet (random init, seed 0), 3k_steps_clipping_rescaling, "
                               f"{B} images/GPU/step sharded over {world} GPU(s), no collectives",
                   "batch_per_gpu": B, "global_batch": B * world, "inference_steps": S, "image_size": size,
                   "hipgraph": not args.no_graph, "concurrent_trajectories": args.streams},
    }
    if box is not None:
        res["box"] = box
    # whole-step roofline numbers the north_star asks for (per GPU): algorithmic activation bytes / flops per image
    per_gpu = value / world
    esz = 2.0 if args.dtype == "f32" else 1.0
    if args.model == "super_small" and size == 256:
        res["step_rooflines"] = {
            "hbm_frac": round(per_gpu * 2 * S * FWD_ACT_MB_PER_IMAGE_BF16 * esz / 1000.0 / PEAK_HBM_GBS, 4),
            "mfma_frac": round(per_gpu * 2 * S * FWD_GFLOP_PER_IMAGE / 1000.0 / PEAK_MFMA_TFLOPS[args.dtype], 4),
            "note": "algorithmic bytes (691.8 MB bf16 act./forward/image) and FLOPs (376 GF/forward/image) x images/s/GPU "
                    "over 8 TB/s HBM and the dense MFMA peak"}
    # The bar that replaces the north-star's ">= 40 % of the HBM roofline" (not reachable by design: the step is not HBM-bound at
    # 16-bit storage and a third of it is the exp-bound d = 8 attention): the SERIAL floor of one UNet forward = each launch at
    # the roofline that binds its kind, one after the other (nothing overlaps inside one trajectory) --
    #   conv / linear FLOPs / dense MFMA peak  +  attention exponentials / v_exp_f32 issue peak  +  bandwidth-only bytes / 8 TB/s
    fl = sum(op.flops for op in runner.plan.ops if op.what != "attn_d8")
    nexp = sum(op.flops for op in runner.plan.ops if op.what == "attn_d8") / 32.0        # 4*B*heads*N^2*8 FLOPs -> B*heads*N^2 exps
    bw = sum(op.bytes for op in runner.plan.ops if op.flops == 0)
    floor_ms = 1e3 * (fl / (PEAK_MFMA_TFLOPS[args.dtype] * 1e12) + nexp / 19.66e12 + bw / (PEAK_HBM_GBS * 1e9))
    fwd_ms = 1e3 * elapsed / args.steps / (2 * S) / args.streams
    res.setdefault("step_rooflines", {}).update({
        "serial_floor_ms": round(floor_ms, 3), "forward_ms": round(fwd_ms, 3), "serial_floor_frac": round(floor_ms / fwd_ms, 4),
        "serial_floor_parts_ms": {"mfma": round(1e3 * fl / (PEAK_MFMA_TFLOPS[args.dtype] * 1e12), 3),
                                  "attention_exp": round(1e3 * nexp / 19.66e12, 3), "hbm_only": round(1e3 * bw / (PEAK_HBM_GBS * 1e9), 3)},
        "serial_floor_note": "per UNet forward at this batch: (conv + linear FLOPs) / dense MFMA peak + attention exponentials / "
                             "19.66 T/s (v_exp_f32 issue at 2.4 GHz) + bytes of the FLOP-free launches / 8 TB/s; forward_ms = "
                             "ms_per_step / (2 x inference_steps)"})

    if rank == 0 and not args.no_roofline:
        # dominant kernel, timed live with HIP events on the launch stream (same inputs, same buffers)
        st = torch.cuda.current_stream(dev).cuda_stream
        prof = runner.plan.profile(runner.x.data_ptr(), runner.temb.data_ptr(), runner.model_out.data_ptr(), st, reps=3)
        torch.cuda.synchronize(dev)
        total_ms = sum(d["ms"] for d in prof.values())
        kind, d = max(prof.items(), key=lambda kv: kv[1]["ms"])
        # an event record between two launches is a full fence (drain + cache write-back): it adds ~5 % to each interval, which
        # rocprofv3's kernel begin/end timestamps do not contain.  The dominant kernel's launch duration is therefore taken
        # from its launches of one forward run back to back between ONE pair of events (same operands, same buffers).
        fenced_ms = d["ms"] / max(d["launches"], 1)
        dom = [op for op in runner.plan.ops if op.what == kind]
        isolated_ms = _back_to_back_ms(dom, st, reps=5) / max(len(dom), 1)
        torch.cuda.synchronize(dev)
        # ... and the figure the roofline is computed from is the launch's duration IN SITU: inside the steady-state forward, at the
        # clock the whole workload holds -- after a short re-warm of the trajectory itself
        one_batch()
        in_situ_ms, in_situ_slots, in_situ_n = _in_situ_launch_ms(runner.plan, kind, st)
        torch.cuda.synchronize(dev)
        d = dict(d, ms=in_situ_ms * len(dom))
        # consistency gate: the dominant kernel's launches + everything else must add up to the forward the timed region measured.
        # Everything else = the event-fenced sum of the other kinds (each interval carries the fence: between 0.90x and 1.0x of it
        # is kernel time) -> the forward implies a band for the dominant launch; a figure outside it (3 % slack) is not printed.
        others_ms = total_ms - fenced_ms * len(dom)
        implied_lo, implied_hi = (fwd_ms - others_ms) / len(dom), (fwd_ms - 0.90 * others_ms) / len(dom)
        in_situ_ok = (not args.no_graph) and args.streams == 1 and 0.97 * implied_lo <= in_situ_ms <= 1.03 * implied_hi
        bound = "mfma" if kind.startswith("conv") else ("mfma" if kind == "attn_d8" else "hbm")
        if bound == "mfma":
            ach, peak, unit = d["flops"] / (d["ms"] * 1e-3) / 1e12, PEAK_MFMA_TFLOPS[args.dtype], "TFLOP/s"
        else:
            ach, peak, unit = d["bytes"] / (d["ms"] * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
        # HBM bytes / pipe occupancy per launch of that kernel come from committed rocprofv3 --pmc passes (counters cannot be read
        # from inside the process): profiles/r4_hbm_traffic.json, r4_mfma_busy.json (r3_* as fallback), written by scripts/collect_profiles.sh keyed
        # by plan kind.  They are only quoted for the workload they were measured on AND while the library sources are the ones
        # they were measured with (sources_sha256); otherwise traffic stays null and the line says why.
        traffic, traffic_src, pipes, pmc_note = None, None, {}, None
        from phendiff_amd._lib import source_hash
        same_workload = B == 32 and args.dtype == "bf16" and size == 256 and args.model == "super_small"
        for stem, field in (("hbm_traffic.json", "traffic"), ("mfma_busy.json", "pipes")):
            j = None
            for rnd in ("r6", "r5", "r4", "r3"):    # the newest collection whose file exists (the source hash decides whether it is quoted)
                fname = f"{rnd}_{stem}"
                try:
                    j = json.load(open(os.path.join(ROOT, "profiles", fname)))
                    break
                except (OSError, ValueError):
                    continue
            if j is None:
                pmc_note = f"profiles/r6_{stem} missing"
                continue
            if j.get("sources_sha256") != source_hash():
                pmc_note = f"profiles/{fname} was measured on other kernel sources (re-run scripts/collect_profiles.sh head)"
                continue
            v = j.get("by_kind", {}).get(kind)
            if not same_workload or v is None:
                pmc_note = pmc_note or f"profiles/{fname} has no entry for {kind} at this workload"
                continue
            if field == "traffic":
                traffic, traffic_src = round(v["hbm_bytes_per_launch"]), f"profiles/{fname}"
            else:
                pipes = {"mfma_util_percent": v["MfmaUtil_percent"], "valu_busy_percent": v["VALUBusy_percent"],
                         "pipes_source": f"profiles/{fname}"}
        if pmc_note:
            pipes["pmc_note"] = pmc_note
        extra = {}
        if kind == "attn_d8":
            # the d=8 attention is bound by the VALU/transcendental issue pipe, not by MFMA or HBM: report
            # exponentials/s against v_exp_f32's 8-cycle issue on 1024 SIMDs at the 2.4 GHz peak clock as well
            nexp = d["flops"] / 32.0                      # flops = 4*B*heads*N^2*8  ->  B*heads*N^2 exps
            extra = {"issue_bound": {"what": "v_exp_f32 issue (64 lanes / 8 cycles / SIMD, 1024 SIMDs, 2.4 GHz)",
                                     "achieved_Texp_per_s": round(nexp / (d["ms"] * 1e-3) / 1e12, 2), "peak_Texp_per_s": 19.66,
                                     "frac": round(nexp / (d["ms"] * 1e-3) / 19.66e12, 4)}}
            # the clock the chip actually holds under this kernel (profiles/r3_clock.json: GRBM_GUI_ACTIVE / 8 / kernel time over
            # 650 launches, and rocm-smi sclk samples, scripts/measure_clock.sh) and the issue cost scripts/micro/exp_variants.hip
            # measures for v_exp_f32 at full occupancy (8.7 cycles, not the nominal 8)
            try:
                clk = json.load(open(os.path.join(ROOT, "profiles", "r3_clock.json")))["N4096"]["ghz_median"]
                peak_meas = 64.0 / 8.7 * 1024 * clk * 1e9
                extra["issue_bound"].update({"measured_clock_ghz": clk, "frac_at_measured_clock": round(nexp / (d["ms"] * 1e-3) / (19.66e12 * clk / 2.4), 4),
                                             "frac_at_measured_clock_and_issue_cost": round(nexp / (d["ms"] * 1e-3) / peak_meas, 4),
                                             "clock_source": "profiles/r3_clock.json"})
            except (OSError, ValueError, KeyError):
                pass
        if kind == "attn_d8":
            # the roofline that BINDS this kernel leads -- the shared VALU / transcendental issue port
            # (16 v_exp_f32 per 32x32 tile) -- and the MFMA figure, which does not bind, is carried second
            ib = extra["issue_bound"]
            extra.update({"mfma_frac": round(ach / peak, 4), "mfma_achieved_tflops": round(ach, 2), "mfma_peak_tflops": peak})
            bound, ach, peak, unit = "valu_transcendental_issue", ib["achieved_Texp_per_s"], ib["peak_Texp_per_s"], "Texp/s"
        res["roofline"] = {**extra, **pipes, "kernel": kind, "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit,
                           "frac": round(ach / peak, 4) if in_situ_ok else None,
                           **({} if in_situ_ok else {"frac_withheld": f"in-situ launch time {in_situ_ms:.4f} ms x {len(dom)} launches does not add up with "
                                                                       f"the forward of the timed region ({fwd_ms:.3f} ms; other kinds {others_ms:.3f} ms event-fenced)"
                                                                       + (": eager / multi-stream run" if args.no_graph or args.streams != 1 else "")}),
                           "traffic": traffic, "traffic_source": traffic_src,
                           "algorithmic_bytes_per_launch": round(d["bytes"] / max(d["launches"], 1)),
                           "launches_per_forward": round(d["launches"]), "avg_launch_ms": round(in_situ_ms, 4),
                           "avg_launch_ms_in_situ_by_launch": [round(v, 4) for v in in_situ_slots], "in_situ_samples": in_situ_n,
                           "avg_launch_ms_isolated": round(isolated_ms, 4),
                           "avg_launch_ms_event_fenced": round(fenced_ms, 4),
                           "avg_launch_ms_implied_by_forward": [round(implied_lo, 4), round(implied_hi, 4)],
                           "in_situ_consistent_with_ms_per_step": bool(in_situ_ok),
                           "share_of_forward": round(fenced_ms * d["launches"] / total_ms, 3),
                           "method": "HIP events on the launch stream. avg_launch_ms (what achieved / frac are computed from): IN SITU -- "
                                     "the plan's launches replayed forward after forward in steady state right after the timed region "
                                     "(same operands and buffers as the captured graph), one event pair per forward around one launch of "
                                     "this kernel, rotating over its launches, first 12 of 72 forwards discarded; it must fall inside "
                                     "avg_launch_ms_implied_by_forward = (ms_per_step / 2S - event-fenced sum of the other kinds [x0.9..1]) "
                                     "/ launches, else frac is withheld; profiles/r6_kernel_stats_b32.csv (rocprofv3 --kernel-trace of "
                                     "the graph replay, first 2 of 4 batches discarded) holds the same figure. avg_launch_ms_isolated: the "
                                     "kernel's launches of one forward back to back between one event pair, 5 reps (boost clock; NOT what "
                                     "the workload sees). per_kernel_* and avg_launch_ms_event_fenced: one event between every two "
                                     "launches of an eager replay, 3 reps (each interval carries the event's fence)",
                           "per_kernel_ms_per_forward": {k: round(v["ms"], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
                           "per_kernel_tflops": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) for k, v in prof.items() if v["ms"] > 0},
                           "per_kernel_gbs": {k: round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) for k, v in prof.items() if v["ms"] > 0}}
    if rank == 0 and world == 1 and args.sweep and not args.no_graph:
        # batch sweep (one warm-up + one timed batch each; the headline batch is the timed region above)
        sweep = {str(B): round(per_gpu, 3)}
        for Bx in (16, 64, 128):
            if Bx == B:
                continue
            del_runner = P.DDIBGraph(pipe, batch_size=Bx, num_inference_steps=S)
            xs, ls = synth_batch(Bx, size, 77)
            xs, ls = xs.to(dev), ls.to(dev)
            del_runner.run(xs, ls, 1 - ls)           # untimed first replay of EVERY point (graph upload, first touch): a sliced batch's
            torch.cuda.synchronize(dev)              # sub-runner is a freshly captured graph too
            t0 = time.perf_counter()
            del_runner.run(xs, ls, 1 - ls)
            torch.cuda.synchronize(dev)
            sweep[str(Bx)] = round(Bx / (time.perf_counter() - t0), 3)
            del del_runner
        res["config"]["sweep"] = {"images_per_s_by_batch": sweep,
                                  "note": f"tensors are addressed with 32-bit byte offsets: one launch plan holds <= {unet.max_batch(size, size)} "
                                          f"images at {size}x{size}; DDIBGraph replays a larger batch as even slices of one captured graph "
                                          f"(B = 128 = 2 x 64)"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(args.model, size, S, state_dict, args.cpu_baseline_seconds)
        res["gpu_over_cpu"] = round(value / res["cpu_baseline"]["value"], 1)
    if rank == 0 and world == 1 and args.side_workloads:
        # every headline field above is final; the headline's plans stay allocated (a few GB of 288) while the children run
        try:
            res["side_workloads"] = run_side_workloads(args.side_budget_s, headline_cpu_baseline=res.get("cpu_baseline"))
        except Exception as e:                                   # never lose the headline line to a side leg
            res["side_workloads"] = {"failed": repr(e)}
    if rank == 0:
        res["diagnostic_env"] = diagnostic_env()
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
