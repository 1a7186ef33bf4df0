"""The overlapped data-parallel step with TWO real ranks on the one GPU a box offers: two child processes on
cuda:0, gloo backend on device tensors, each running `UNetTrainer.step(..., overlap=True)` / `SDUNetTrainer.step` on its half of
a batch at a different pace -- see tests/two_rank_overlap_worker.py for what each rank asserts (gradient == torch.autograd of the
oracle on the concatenated batch; parameters bit-identical across ranks after 3 steps from different initial weights; class table
untouched by the unconditional step).  Reference: train.py:62-74,311-326 (DDP wrap), utils_training.py:436 (backward)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run_two_ranks(which, timeout=420):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=os.pathsep.join([os.path.dirname(HERE), HERE]))
        env.pop("PYTEST_CURRENT_TEST", None)
        # fresh interpreters started as CHILD processes (never an exec of this GPU-initialised one)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "two_rank_overlap_worker.py"), which], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed (rc {p.returncode}):\n{out[-3000:]}"
        assert f"two_rank_overlap_worker {which} rank {rank}: OK" in out, out[-2000:]


def test_two_rank_overlapped_step_pixel_unet():
    _run_two_ranks("pixel")


def test_two_rank_overlapped_step_sd_unet():
    _run_two_ranks("sd")


def test_two_rank_attention_fine_tuning_exchanges_trainable_runs_only():
    """--attention_fine_tuning (train.py:201-220) with two ranks: buckets are cut inside the runs of trainable parameters."""
    _run_two_ranks("pixel_frozen")
