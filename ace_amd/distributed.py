"""The slice of the `Distributed` facade the hot path touches (fme/core/distributed/distributed.py:28-519,
non_distributed.py:29-140, torch_distributed.py:27-204): one process per GPU (torchrun convention:
RANK / LOCAL_RANK / WORLD_SIZE), backend "nccl" (= RCCL over xGMI on ROCm) on GPU or "gloo" on CPU.

The rollout shards naturally: ensemble members / initial conditions never interact inside a step
(reference: round-robin IC sharding, fme/ace/data_loading/inference.py:291-298).  The ONLY collective on the
path is the ensemble-mean diagnostic: sum-all-reduce of the denormalised output state divided by the number of
ranks (`reduce_mean`, torch_distributed.py:130-132), at diagnostic cadence."""

import os
from typing import Optional

import torch
import torch.distributed as dist

from .sht import InverseRealSHT, RealSHT


class DummyWrapper(torch.nn.Module):
    """non_distributed.py:15-28: gives state_dict keys the same 'module.' prefix as DDP."""

    def __init__(self, module: torch.nn.Module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


class Distributed:
    _instance: Optional["Distributed"] = None

    def __init__(self, backend: Optional[str] = None):
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self._owns_group = False
        if self.world_size > 1 and not dist.is_initialized():
            use_gpu = torch.cuda.is_available() and os.environ.get("FME_FORCE_CPU", "0") != "1"
            # ACE_DIST_BACKEND=gloo: the N > 1 path on device tensors without RCCL - a pre-flight of bench.py --gpus N on a box with
            # fewer devices than ranks (ranks then share devices: local rank modulo the device count; RCCL refuses that)
            backend = backend or os.environ.get("ACE_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
            if use_gpu:
                torch.cuda.set_device(self.local_rank % torch.cuda.device_count())
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            kw = {}
            if use_gpu and backend == "nccl":
                kw["device_id"] = torch.device("cuda", self.local_rank)
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world_size, **kw)
            self._owns_group = True

    @classmethod
    def get_instance(cls) -> "Distributed":
        if cls._instance is None:
            cls._instance = cls()
        return cls._instance

    @classmethod
    def reset(cls):
        if cls._instance is not None:
            cls._instance.shutdown()
        cls._instance = None

    def is_distributed(self) -> bool:
        return self.world_size > 1 and dist.is_initialized()

    def is_root(self) -> bool:
        return self.rank == 0

    # -- partitioning: member g <-> rank g % world_size (inference.py:296-298)
    def owns_member(self, i_member: int) -> bool:
        return i_member % self.world_size == self.rank

    def local_members(self, n_members: int):
        return [i for i in range(n_members) if self.owns_member(i)]

    # -- reductions (torch_distributed.py:113-160)
    def reduce_sum(self, tensor: torch.Tensor) -> torch.Tensor:
        if self.is_distributed():
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
        return tensor

    def reduce_mean(self, tensor: torch.Tensor) -> torch.Tensor:
        if self.is_distributed():
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
            tensor /= self.world_size
        return tensor

    def reduce_max(self, tensor: torch.Tensor) -> torch.Tensor:
        if self.is_distributed():
            dist.all_reduce(tensor, op=dist.ReduceOp.MAX)
        return tensor

    def barrier(self):
        if self.is_distributed():
            dist.barrier()

    # -- module / transform factories (distributed.py:443-447, non_distributed.py)
    def wrap_module(self, module: torch.nn.Module) -> torch.nn.Module:
        return DummyWrapper(module)

    def get_sht(self, nlat, nlon, lmax=None, mmax=None, grid="legendre-gauss"):
        return RealSHT(nlat, nlon, lmax=lmax, mmax=mmax, grid=grid)

    def get_isht(self, nlat, nlon, lmax=None, mmax=None, grid="legendre-gauss"):
        return InverseRealSHT(nlat, nlon, lmax=lmax, mmax=mmax, grid=grid)

    def get_local_slices(self, tensor_shape):
        return tuple(slice(None) for _ in tensor_shape)

    def shutdown(self):
        if self._owns_group and dist.is_initialized():
            dist.destroy_process_group()
        self._owns_group = False


class AsyncEnsembleMean:
    """The ensemble-mean diagnostic OFF the step stream ("on a side stream, at diagnostic cadence").

    `submit(fields)` records an event on the caller's stream, and on a side stream - after that event - stacks the fields into a
    reduction buffer owned by this object, all-reduces it over RCCL (the process group's collective is ordered after the side
    stream, not after the step stream) and divides by the number of ranks.  The step stream never waits: the next step's
    kernels are enqueued while the reduce is in flight; the producer may overwrite its fields only after `wait()` /
    `result()` (or after its own stream has waited on `done`).  On CPU tensors (gloo) the same calls run synchronously.
    Semantics: gen.mean(dim=members) then reduce_mean (fme/ace/aggregator/one_step/ensemble.py:93-112,299;
    fme/core/distributed/torch_distributed.py:130-132)."""

    def __init__(self, shape, device, distributed: Optional[Distributed] = None):
        self.dist = distributed or Distributed.get_instance()
        self.device = torch.device(device)
        self.buf = torch.zeros(tuple(shape), device=self.device)
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(self.device) if self.cuda else None
        self.done = torch.cuda.Event(enable_timing=True) if self.cuda else None
        self.begin = torch.cuda.Event(enable_timing=True) if self.cuda else None
        self._pending = False

    def submit(self, fields) -> None:
        """fields: a sequence of equally shaped tensors (one per output name) or one tensor of the buffer's shape, produced on the
        CURRENT stream"""
        if not self.cuda:
            self.buf.copy_(fields if torch.is_tensor(fields) else torch.stack(list(fields)))
            self.dist.reduce_mean(self.buf)
            self._pending = True
            return
        fields = fields if torch.is_tensor(fields) else list(fields)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            # the inputs are read on the side stream: tell the caching allocator, so that a temporary the caller drops right
            # after submit() (2.0 * state) is not handed to step-stream kernels before the stack / copy below has run
            for t in ([fields] if torch.is_tensor(fields) else fields):
                t.record_stream(self.stream)
            if torch.is_tensor(fields):
                self.buf.copy_(fields)
            else:
                torch.stack(fields, out=self.buf)
            self.begin.record(self.stream)
            self.dist.reduce_mean(self.buf)
            self.done.record(self.stream)
        self._pending = True

    def wait(self) -> None:
        """make the CURRENT stream wait for the last submitted reduction (no host synchronisation)"""
        if self.cuda and self._pending:
            torch.cuda.current_stream(self.device).wait_event(self.done)

    def result(self) -> torch.Tensor:
        """the reduced mean, after a host synchronisation on the side stream"""
        if self.cuda and self._pending:
            self.done.synchronize()
        return self.buf

    def last_allreduce_ms(self) -> Optional[float]:
        """device time of the last all-reduce + scale on the side stream (None before the first one / on CPU)"""
        if not (self.cuda and self._pending):
            return None
        self.done.synchronize()
        return float(self.begin.elapsed_time(self.done))


class EnsembleMean:
    """Running ensemble-mean diagnostic over members that live on different ranks
    (fme/ace/aggregator/one_step/ensemble.py:93-112,299): mean over local members, then reduce_mean."""

    def __init__(self, distributed: Optional[Distributed] = None):
        self.dist = distributed or Distributed.get_instance()

    def __call__(self, local_members: torch.Tensor, member_dim: int = 0) -> torch.Tensor:
        local_mean = local_members.mean(dim=member_dim).contiguous()
        return self.dist.reduce_mean(local_mean)
