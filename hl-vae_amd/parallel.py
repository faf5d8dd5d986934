"""Data parallelism for the ELBO step: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests and in the two-ranks-on-one-GPU rehearsal).

The reference has no distributed code at all.  The semantics to keep are its single-process ones:
batches are WHOLE subjects (the GP prior needs each subject's T x T block, utils.py:77-97; elbo_functions.py:243-252), the loss
carries P / P_batch with the GLOBAL P_batch and ONE optimiser step is taken on it (training.py:121-128), and the rows of a
batch are coupled through the batch statistics (HL_VAE/utils.py:105-108, 126-132).

Exchange steps of one training step, world = N ranks:

  1. all-reduce of the masked column-sum partials of the real / pos variables (125 kB fp64 for D4): every rank normalises
     with the statistics of the global batch.  It belongs to the input stage, which runs one batch AHEAD on a side stream;
  2. reduce-scatter of the dense part of the fp32 gradient arena, in two contiguous slices: y_layer's weight gradient
     (55 % of the arena, final first: it overlaps the rest of the backward pass), then the other matrices.  Each rank receives
     the SUM over ranks of 1 / N of every slice;
  3. all-reduce of the small region of the arena (head parameters, biases: 0.13 MB) -- replicated Adam on it;
  4. Adam on this rank's slices only (`ShardPlan`): 28 B per parameter of optimiser traffic / N, and a bf16 copy of the
     updated values;
  5. all-gather of the bf16 copies (2 B per parameter instead of the 4 B an all-reduce would have moved back), from which
     every rank rebuilds the padded shadows its MFMA kernels read.  The fp32 masters of the other ranks' slices are NOT
     refreshed per step (nothing on the step reads them); `ShardedState.sync_masters` gathers them on demand (state_dict).
  6. (GP prior) the per-subject partial sums of the KL, see elbo_functions.py in this package.

Bytes on the wire per rank and step for D4 (P_n = 5.88 M dense parameters) at N = 8: reduce-scatter 7/8 x 23.5 MB out,
all-gather 7/8 x 11.8 MB out = 30.9 MB, against 41.2 MB for a ring all-reduce of the fp32 arena; the optimiser pass
shrinks from 183 MB to 23 MB of HBM traffic per GPU.  The same code runs at N = 1 (the collectives are skipped).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def _ru(v: int, m: int) -> int:
    return (v + m - 1) // m * m


@dataclass
class ShardSlice:
    """one contiguous arena range [lo, hi) cut into `world` equal chunks of `chunk` elements (the last ones clipped at hi)"""
    lo: int
    hi: int
    chunk: int
    which: int                    # matrices of this slice (bit mask of hlvae_shadows_from_bf16)

    def own(self, rank: int) -> Tuple[int, int]:
        a = min(self.hi, self.lo + rank * self.chunk)
        return a, max(a, min(self.hi, self.lo + (rank + 1) * self.chunk))


class ShardPlan:
    """Equal flat slices of the dense arena region, one per rank.  ``ranges`` = [(lo, hi, which), ...] in the order their
    gradients become final.  Chunks are multiples of ``align`` elements (16-byte vector accesses, 128-byte lines for fp32)."""

    def __init__(self, ranges: Sequence[Tuple[int, int, int]], world: int, rank: int, align: int = 32):
        if not (0 <= rank < world):
            raise ValueError(f"rank {rank} outside world {world}")
        self.world, self.rank, self.align = world, rank, align
        self.slices: List[ShardSlice] = []
        for lo, hi, which in ranges:
            if lo % align or hi < lo:
                raise ValueError(f"slice [{lo}, {hi}) must start at a multiple of {align}")
            self.slices.append(ShardSlice(lo, hi, _ru(-(-(hi - lo) // world), align) if hi > lo else 0, which))

    @property
    def pad(self) -> int:
        """elements a slice's padded extent (world * chunk) can reach past its end: tail slack the arenas need"""
        return max((s.lo + self.world * s.chunk - s.hi for s in self.slices), default=0)

    def owned(self) -> List[Tuple[int, int]]:
        return [s.own(self.rank) for s in self.slices]

    def owner_of(self, i: int) -> int:
        for s in self.slices:
            if s.lo <= i < s.hi:
                return (i - s.lo) // s.chunk
        raise IndexError(i)


class DataParallel:
    """Collectives of the step.  RCCL has native reduce-scatter / all-gather; gloo (CPU tests, rehearsals) does not have
    reduce-scatter: the same result is produced with an all-reduce of a copy + a local slice."""

    def __init__(self, group=None):
        self.group = group if group is not None else dist.group.WORLD
        self.world = dist.get_world_size(self.group)
        self.rank = dist.get_rank(self.group)
        self.backend = dist.get_backend(self.group)
        self.native = self.backend == "nccl"

    @classmethod
    def single(cls) -> "DataParallel":
        """world of one without a process group: the data-parallel code path with every collective skipped"""
        self = cls.__new__(cls)
        self.group, self.world, self.rank, self.backend, self.native = None, 1, 0, "none", False
        return self

    # ---- small replicated exchanges -----------------------------------------------------------------------------
    def allreduce_stats(self, sums: torch.Tensor):
        if self.world > 1:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group)

    def allreduce_(self, t: torch.Tensor):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    allreduce_grads = allreduce_

    def allreduce_async(self, t: torch.Tensor):
        """starts the collective on the backend's own stream (it first waits for the work already queued on the current
        stream); ``.wait()`` on the returned handle makes the current stream wait for the result"""
        if self.world == 1:
            return _Done()
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def broadcast_(self, t: torch.Tensor, src: int = 0):
        """replicated state (GP hyper-parameters, inducing points, m, H) starts identical on every rank"""
        if self.world > 1:
            dist.broadcast(t, src=dist.get_global_rank(self.group, src) if self.group is not dist.group.WORLD else src, group=self.group)
        return t

    # ---- the sharded optimiser's two collectives ----------------------------------------------------------------
    def reduce_scatter(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        """out [chunk] = sum over ranks of inp[rank * chunk : (rank + 1) * chunk]; inp has world * chunk elements."""
        assert inp.numel() == self.world * out.numel(), (inp.numel(), out.numel(), self.world)
        if self.world == 1:
            out.copy_(inp)
            return _Done()
        if self.native:
            w = dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            return w if async_op else _Done()
        tmp = inp.clone()
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group)
        out.copy_(tmp[self.rank * out.numel():(self.rank + 1) * out.numel()])
        return _Done()

    def all_gather(self, full: torch.Tensor, mine: torch.Tensor, async_op: bool = False):
        """full [world * chunk] = concatenation of every rank's ``mine`` [chunk] (``mine`` may be the matching view of ``full``)."""
        assert full.numel() == self.world * mine.numel()
        if self.world == 1:
            if full.data_ptr() != mine.data_ptr():
                full.copy_(mine)
            return _Done()
        if self.native:
            w = dist.all_gather_into_tensor(full, mine, group=self.group, async_op=async_op)
            return w if async_op else _Done()
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine.clone(), group=self.group)
        full.copy_(torch.cat(parts))
        return _Done()


class _Done:
    def wait(self):
        return True


class ShardedState:
    """Buffers and collectives of the sharded optimiser step for one flat arena (device independent: the GPU trainer plugs in
    the HIP kernels, the CPU tests an fp64 restatement).

        grads  G [arena + pad] fp32          ->  reduce_scatter(slice k)  ->  gsh[k] [chunk_k]  (sum over ranks)
        adam(k, lo, n, gsh[k], pb[k] shard)      on the owned range
        pb[k] [world * chunk_k] bf16         <-  all_gather(slice k)
    """

    def __init__(self, dp: DataParallel, plan: ShardPlan, device, grad_dtype=torch.float32, copy_dtype=torch.bfloat16, slack=0):
        """slack: extra zero elements behind every gathered copy (a consumer that reads a padded matrix straight out of the copy)"""
        self.dp, self.plan = dp, plan
        self.gsh = [torch.zeros(max(s.chunk, 1), dtype=grad_dtype, device=device) for s in plan.slices]
        self.pb = [torch.zeros(max(plan.world * s.chunk, 1) + slack, dtype=copy_dtype, device=device) for s in plan.slices]

    def reduce_scatter_slice(self, k: int, G: torch.Tensor, async_op: bool = False):
        s = self.plan.slices[k]
        if s.chunk == 0:
            return _Done()
        return self.dp.reduce_scatter(self.gsh[k][:s.chunk], G[s.lo:s.lo + self.plan.world * s.chunk], async_op=async_op)

    def all_gather_slice(self, k: int, async_op: bool = False):
        s = self.plan.slices[k]
        if s.chunk == 0:
            return _Done()
        r = self.plan.rank
        return self.dp.all_gather(self.pb[k][:self.plan.world * s.chunk], self.pb[k][r * s.chunk:(r + 1) * s.chunk], async_op=async_op)

    def own_copy_view(self, k: int) -> torch.Tensor:
        s, r = self.plan.slices[k], self.plan.rank
        return self.pb[k][r * s.chunk:(r + 1) * s.chunk]

    def sync_masters(self, P: torch.Tensor):
        """gather every rank's fp32 master slices into the full arena ``P`` (state_dict, checkpoints, parity checks)"""
        if self.plan.world == 1:
            return
        for s in self.plan.slices:
            if s.chunk == 0:
                continue
            lo, hi = s.own(self.plan.rank)
            mine = torch.zeros(s.chunk, dtype=P.dtype, device=P.device)
            mine[:hi - lo].copy_(P[lo:hi])
            full = torch.empty(self.plan.world * s.chunk, dtype=P.dtype, device=P.device)
            self.dp.all_gather(full, mine)
            P[s.lo:s.hi].copy_(full[:s.hi - s.lo])
