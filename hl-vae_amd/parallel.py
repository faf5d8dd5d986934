"""Data parallelism for the ELBO step: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

The path shards over WHOLE subjects (the reference batches by subject because the GP prior needs each
subject's T x T block, utils.py:77-97; elbo_functions.py:243-252).  Exchange steps per training step:

  1. masked column sums of the real / pos variables  (3 x n_stat fp64, a few kB) so that every rank
     normalises with the statistics of the GLOBAL batch, exactly as one process would
     (HL_VAE/utils.py:105-108, 126-132 couple the rows of a batch);
  2. ONE sum all-reduce of the flat fp32 gradient arena.  The loss is scaled by P / P_batch with the global
     P_batch (training.py:121-122), so the sum over ranks IS the single-process gradient;
  3. (GP prior) the per-subject partial sums of the KL, see elbo_functions.py in this package.

The reference has no distributed code at all; this is new design, not a translation.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, group=None):
        self.group = group if group is not None else dist.group.WORLD
        self.world = dist.get_world_size(self.group)
        self.rank = dist.get_rank(self.group)

    def allreduce_stats(self, sums: torch.Tensor):
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group)

    def allreduce_grads(self, grad_arena: torch.Tensor):
        dist.all_reduce(grad_arena, op=dist.ReduceOp.SUM, group=self.group)

    def allreduce_async(self, t: torch.Tensor):
        """starts the collective on the backend's own stream (it first waits for the work already queued on the current
        stream); ``.wait()`` on the returned handle makes the current stream wait for the result"""
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def broadcast_(self, t: torch.Tensor, src: int = 0):
        """replicated state (GP hyper-parameters, inducing points, m, H) starts identical on every rank"""
        dist.broadcast(t, src=dist.get_global_rank(self.group, src) if self.group is not dist.group.WORLD else src, group=self.group)
        return t

    def allreduce_(self, t: torch.Tensor):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t
