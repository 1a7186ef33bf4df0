"""N > 1 path on CPU: world_size-2 gloo process group.  Inference shards independent batches across ranks with no
data-path collective (utils_Img2Img.py:316-317); the only communication is bench.py's barrier + MAX-reduce of the
elapsed time.  Both are exercised here with the product's own sharding function."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import phendiff_amd as P


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_batches, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = P.shard_batches(num_batches, rank, world)
    # "process" each batch: a deterministic per-batch result, as independent images would give
    results = {b: float(torch.manual_seed(1000 + b).initial_seed()) for b in mine}
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, results))
    dist.barrier()
    t = torch.tensor([0.5 + rank], dtype=torch.float64)      # bench.py: MAX over ranks of the elapsed time
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        out.put((gathered, float(t)))
    dist.destroy_process_group()


def test_two_rank_sharding_covers_every_batch_once_per_round():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, out)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, tmax = out.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (b0, r0), (b1, r1) = gathered
    assert b0 == [0, 2, 4, 6] and b1 == [1, 3, 5, 0]          # rank r takes r, r+2, ...; the tail wraps around
    assert len(b0) == len(b1)
    assert set(b0) | set(b1) == set(range(7))
    assert r1[0] == r0[0]                                        # the duplicated batch gives the same result (dedupe on save)
    assert tmax == 1.5


# ---- bench.py's start-up self-test of the exchange (VERDICT r4 next 2 / ADVICE r4): the native legs' control flow on two gloo ranks ----
SELFTEST_MODES = ("default", "timeout", "id_error", "init_error", "leg_error", "ok")


def _selftest_worker(rank, world, port, out):
    import json
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    res = {}
    for mode in SELFTEST_MODES:          # one process group for all of them: each must leave it in step for the next

        class Stub(bench._StubComm):
            spec = (1, 4.0) if mode == "timeout" else (-1, 0.0)

            def __init__(self, rank_, world_, cid, device=None, mode=mode):
                self.mode = mode
                if mode == "init_error" and rank_ == 0:
                    raise RuntimeError("pd_comm_init: RCCL error 2")
                super().__init__(rank_, world_, cid, device)

            @staticmethod
            def unique_id(mode=mode):
                if mode == "id_error":
                    raise RuntimeError("librccl.so could not be loaded")
                return bench._StubComm.unique_id()

            def allreduce_(self, flat, mean=True, algo=1, stream=None):
                super().allreduce_(flat, mean, algo, stream)      # (the stub's data path is the torch group: both ranks enter it)
                if self.mode == "leg_error" and algo == 1 and self.rank == 1:
                    raise RuntimeError("pd_allreduce_bucket: RCCL error 5")
                return flat

        t0 = time.time()
        st = bench.comm_selftest(dist, dev, nbytes=1 << 20, native_timeout_s=1.0, native=(mode != "default"), comm_cls=None if mode == "default" else Stub)
        took = time.time() - t0
        # the process group is still in step on every rank: the timed region's collectives (bench.reduce_elapsed) go through, and rank 0's line prints
        bench._SELFTEST["result"] = st
        elapsed, ranks = bench.reduce_elapsed(dist, 1.0 + rank, dev, 10)
        res[mode] = (st, elapsed, json.dumps({"value": 1.0, **ranks}), took)
    dist.barrier()
    out.put((rank, res))
    dist.destroy_process_group()


def test_bench_selftest_control_flow_on_two_ranks():
    import json
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_selftest_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(out.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, res in got:
        for mode in SELFTEST_MODES:
            st, elapsed, line, took = res[mode]
            assert elapsed == 2.0, mode                       # MAX over the ranks' clocks: the group's collectives are still in step
            assert st["exact"]["torch"] is True and st["busbw_GBs"]["torch"] > 0, mode
            assert '"allreduce_selftest"' in line
        # the C-ABI legs are opt-in: the driver's default command runs the torch leg only
        assert res["default"][0]["native"].startswith("not run")
        # a rank stuck in pd_comm_init: every rank reports "timeout", nobody waits for the sleeper, and the line still prints
        st, _, line, took = res["timeout"]
        assert st["native"] == "timeout" and "rs_ag" not in st["exact"] and took < 3.5
        assert json.loads(line)["allreduce_selftest"]["native"] == "timeout"
        # failures anywhere are agreed by all ranks
        assert res["id_error"][0]["native"].startswith("no communicator id")
        assert res["init_error"][0]["native"] == "error"
        st = res["leg_error"][0]
        assert st["exact"]["rccl_allreduce"] is True and st["busbw_GBs"]["rs_ag"] is None
        st = res["ok"][0]
        assert st["native"] == "ran" and st["exact"]["rs_ag"] is True and st["rccl_world_size"] == 2 and st["rccl_rank_matches"]
