"""bench.py's N > 1 control flow, rehearsed with two ranks on the one GPU of the test box (gloo instead of RCCL, which refuses two
ranks per device): launcher command, process group, barriers around the timed region, MAX over the ranks' clocks via all_gather,
one JSON line from rank 0 with whole-job throughput -- for the headline workload and the data-parallel training workload (whose
overlapped gradient all-reduce then really runs between two ranks).  The driver's 8-GPU run goes through exactly this code with
backend nccl."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, **env_extra):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-roofline", "--no-sweep"] + extra
    env = dict(os.environ, PD_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    env.pop("PYTEST_CURRENT_TEST", None)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                  # ONE line, from rank 0
    return json.loads(lines[0])


def test_bench_two_ranks_img2img():
    j = _run(["--batch", "4", "--size", "64", "--inference-steps", "3"])
    assert j["n_gpus"] == 2 and j["rccl_world_size"] == 2 and len(j["per_rank_units_per_s"]) == 2
    assert j["scaling"] == "weak" and j["config"]["global_batch"] == 8 and "side_workloads" not in j
    # whole-job value = units of all ranks / the slowest rank's time
    assert abs(j["value"] - 2 * 4 * j["steps"] / (j["ms_per_step"] * j["steps"] / 1e3)) < 1e-2 * j["value"]
    assert j["value"] <= sum(j["per_rank_units_per_s"]) * 1.001 and j["diagnostic_env"].get("PD_BENCH_REHEARSAL") == "1"
    # the start-up self-test of the exchange: 64 MB all-reduced and checked bit for bit against the analytic sum
    st = j["allreduce_selftest"]
    assert st["world"] == 2 and st["bytes"] >= 60 << 20 and st["exact"]["torch"] is True and st["busbw_GBs"]["torch"] > 0
    # the C-ABI legs are opt-in: the driver's default command runs the torch leg only
    assert st["native"].startswith("not run") and j["rccl_world_size_source"].startswith("torch.distributed")


def test_bench_two_ranks_rank_stuck_in_comm_init_still_prints_the_line():
    """a rank that does not come back from pd_comm_init (a stub sleeping past the deadline on rank 1) costs the run
    nothing -- every rank agrees on "timeout" from its main thread, the timed region runs, rank 0 prints the line."""
    j = _run(["--batch", "2", "--size", "32", "--inference-steps", "2"], PD_BENCH_SELFTEST_STUB="1:20", PD_BENCH_NATIVE_TIMEOUT_S="2")
    st = j["allreduce_selftest"]
    assert st["native"] == "timeout" and st["exact"]["torch"] is True and "rs_ag" not in st["exact"]
    assert j["n_gpus"] == 2 and j["value"] > 0 and len(j["per_rank_units_per_s"]) == 2


def test_bench_two_ranks_training():
    j = _run(["--workload", "train", "--batch", "8", "--size", "32"])
    assert j["n_gpus"] == 2 and j["rccl_world_size"] == 2 and j["config"]["global_batch"] == 16
    assert j["config"]["final_loss"] == j["config"]["final_loss"] and j["value"] > 0
    # what would explain a bent scaling curve: overlap fraction of the exchange and the step
    # with the exchange left out, per rank
    dp = j["data_parallel"]
    assert 0.0 <= dp["allreduce_overlap_frac"] <= 1.0 and dp["allreduce"]["buckets"] >= 1 and dp["allreduce"]["bytes"] == 4 * 15_725_443
    assert dp["allreduce"]["comm_ms"] >= dp["allreduce"]["exposed_ms"] >= 0
    assert 0 < dp["step_ms_no_comm"] <= dp["step_ms_with_comm"] * 1.5
    assert len(dp["per_rank"]["step_ms_no_comm"]) == 2 and j["allreduce_selftest"]["exact"]["torch"] is True


def test_bench_two_ranks_sd_training_reports_the_exchange():
    j = _run(["--workload", "sd_train", "--batch", "2", "--size", "16"])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 4 and j["value"] > 0
    dp = j["data_parallel"]
    assert 0.0 <= dp["allreduce_overlap_frac"] <= 1.0 and dp["allreduce"]["bytes"] > 3.4e9 and dp["step_ms_no_comm"] > 0
