"""bench.py launched bare with --gpus N (the driver's form) becomes the launcher: N fresh ranks under torch.distributed.run on
127.0.0.1 -- the reference's `accelerate launch --multi_gpu --num_processes N` (launch_script_DDIM.sh:19-34)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [[], ["--workload", "train"], ["--workload", "sd_img2img", "--batch", "4"], ["--workload", "sd_train"]])
def test_bare_multi_gpu_invocation_builds_the_launch_command(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PD_BENCH_PRINT_LAUNCH"] = "1"
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "1"] + extra
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    cmd = json.loads(r.stdout.strip().splitlines()[-1])
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                     # the ranks see exactly the user's flags


def test_more_ranks_than_devices_is_refused_before_any_gpu_work():
    if torch.cuda.device_count() >= 8:
        pytest.skip("needs a box with fewer than 8 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PD_BENCH_PRINT_LAUNCH")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "exposes" in r.stderr


def test_reduce_elapsed_reports_every_rank():
    sys.path.insert(0, ROOT)
    import bench
    el, info = bench.reduce_elapsed(None, 2.0, "cpu", 64)
    assert el == 2.0 and info == {"rccl_world_size": 1, "per_rank_units_per_s": [32.0]}


def test_a_rank_whose_process_group_init_fails_exits_non_zero_with_one_line():
    """a rank that cannot join the process group prints ONE line and exits 3 (no traceback wall, no hang) -- before
    anything touched the GPU in that process.  The failure is simulated (PD_BENCH_FAIL_INIT=<rank>); runs on the CPU box."""
    env = {k: v for k, v in os.environ.items() if k not in ("PD_BENCH_PRINT_LAUNCH", "PD_BENCH_REHEARSAL")}
    env.update(WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", PD_BENCH_FAIL_INIT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stderr.splitlines() if ln.startswith("bench.py:")]
    assert r.returncode == 3 and len(lines) == 1 and "rank 1/2" in lines[0] and "init_process_group failed" in lines[0], r.stderr[-1500:]
    assert "Traceback" not in r.stderr
