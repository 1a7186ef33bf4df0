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
