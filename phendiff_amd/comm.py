"""The data-parallel gradient exchange through the C ABI (`pd_comm_*`, `csrc/comm_rccl.hip`): RCCL driven directly, one communicator per
process / GPU -- what `DistributedDataParallel` does under `accelerator.backward(loss)` in the reference (train.py:311-326,
utils_training.py:436).  The trainers use `torch.distributed` by default (PyTorch is plumbing in this repo); `NativeComm` is the same
exchange for a host program that has no torch process group, or wants the reduce-scatter + all-gather form (`algo=1`) SURVEY 5.8 asks
for on the 7 point-to-point xGMI links.  The 128-byte id is drawn by rank 0 and shipped by the caller (here: a torch.distributed
broadcast when a process group exists)."""
import ctypes as C
from typing import Optional

import torch

from . import _lib as L


class NativeComm:
    def __init__(self, rank: int, world: int, comm_id: Optional[bytes] = None, device=None):
        """``comm_id``: the 128 bytes rank 0 got from :meth:`unique_id` (required unless world == 1)."""
        self.lib = L.lib()
        self.rank, self.world = int(rank), int(world)
        if device is not None:
            torch.cuda.set_device(device)
        if comm_id is None:
            if world != 1:
                raise ValueError("NativeComm: every rank needs rank 0's id (NativeComm.unique_id()) when world > 1")
            comm_id = self.unique_id()
        if len(comm_id) != 128:
            raise ValueError("NativeComm: the communicator id is 128 bytes")
        cid = L.CommId()
        C.memmove(C.byref(cid), comm_id, 128)
        self._comm = C.c_void_p()
        L.check(self.lib.pd_comm_init(C.byref(cid), self.rank, self.world, C.byref(self._comm)), "pd_comm_init")

    @staticmethod
    def unique_id() -> bytes:
        cid = L.CommId()
        L.check(L.lib().pd_comm_unique_id(C.byref(cid)), "pd_comm_unique_id")
        return C.string_at(C.byref(cid), 128)

    @classmethod
    def from_process_group(cls, group=None, device=None):
        """Rank 0 draws the id, the torch process group carries it to the others (any backend)."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(rank, world, box[0], device)

    def allreduce_(self, flat: torch.Tensor, mean: bool = True, algo: int = 1, stream=None) -> torch.Tensor:
        """In-place sum (mean) over the ranks of one contiguous fp32 device tensor, asynchronously on ``stream`` (default: current)."""
        if flat.dtype != torch.float32 or not flat.is_cuda or not flat.is_contiguous():
            raise L.PhenDiffHipError("NativeComm.allreduce_: a contiguous fp32 device tensor")
        st = (stream or torch.cuda.current_stream(flat.device)).cuda_stream
        L.check(self.lib.pd_allreduce_bucket(self._comm, flat.data_ptr(), flat.numel(), int(mean), int(algo), st), "pd_allreduce_bucket")
        return flat

    def query(self):
        """(rank, world) as the communicator itself reports them (ncclCommUserRank / ncclCommCount)."""
        r, w = C.c_int(-1), C.c_int(-1)
        L.check(self.lib.pd_comm_query(self._comm, C.byref(r), C.byref(w)), "pd_comm_query")
        return r.value, w.value

    def close(self):
        if getattr(self, "_comm", None):
            self.lib.pd_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
