"""
Data-parallel plumbing for the scattering path.

Graphs of a batch are independent diagonal blocks of P and Q, so the scattering
itself shards over ranks with NO exchange: each rank takes a contiguous range
of graphs.  The only collective of a training step is one all-reduce of the
gradients of the (small, ~1e5 parameter) heads that consume the coefficients;
the reference gets it from Accelerate/DDP (training/train.py:408, :483-492).
Here it is one flat fp32 bucket: at ~0.4 MB the all-reduce is latency-bound on
xGMI, so bucketing/overlap would only add launches.

One process per GPU; backend "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU
tests.  Rendezvous comes from the environment (RANK / WORLD_SIZE / MASTER_*).
"""
import os
from typing import Iterable, List, Tuple
import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def graph_shard(n_graphs: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous [start, stop) of graphs for `rank`; sizes differ by at most one (split_batches-style,
    config/train_config.py:132)."""
    base, rem = divmod(n_graphs, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def node_range_of_graphs(graph_ptr, g0: int, g1: int) -> Tuple[int, int]:
    return int(graph_ptr[g0]), int(graph_ptr[g1])


def max_over_ranks(seconds: float, device=None) -> float:
    """the slowest rank's time (what bench.py reports), identity when not distributed."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if dist.get_backend() != "gloo" else None)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class FlatGradBucket:
    """
    One flat fp32 buffer aliasing the .grad of every parameter: backward writes
    into it, `allreduce_mean()` is a single collective, the optimizer reads the
    same memory.  (What DDP's bucket view does, without the hooks: the heads are
    tiny and there is nothing to overlap with.)
    """

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, n = self.params[0].device, sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def allreduce_mean(self):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if self.flat.is_cuda and dist.get_backend() == "gloo":
                # rehearsal / tests (ranks sharing a GPU over gloo): stage through the host; the product path is RCCL
                h = self.flat.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                self.flat.copy_(h)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.mul_(1.0 / dist.get_world_size())
