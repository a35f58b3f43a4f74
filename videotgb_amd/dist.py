"""Multi-GPU side of the path: one process per GPU, clips sharded with no data-path collective.

Inference: the clip list is cut into contiguous chunks of ceil(len / n) clips, chunk k
goes to rank k (eval/inference.py:21-29, driven per GPU by eval/scripts/run_qa_*.sh:16-48); ranks never
exchange data, results are concatenated in rank order.  Training (config 5): the only exchange is
the sum all-reduce of the trainable gradients (LoRA: 16.8 MB) once per optimizer step -- one flat
pre-allocated fp32 bucket, torch.distributed (backend "nccl" == RCCL over xGMI on the MI355X node,
"gloo" in the CPU tests).
"""
from __future__ import annotations

import math
from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


def split_list(lst: Sequence, n: int) -> List[Sequence]:
    """eval/inference.py:21-24: n (roughly) equal contiguous chunks of ceil(len / n) elements."""
    chunk = math.ceil(len(lst) / n)
    return [lst[i:i + chunk] for i in range(0, len(lst), chunk)]


def get_chunk(lst: Sequence, n: int, k: int) -> Sequence:
    """eval/inference.py:27-29.  A rank past the last chunk gets nothing (the reference would raise IndexError)."""
    chunks = split_list(lst, n)
    return chunks[k] if k < len(chunks) else lst[:0]


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def max_over_ranks(seconds: float, device="cpu") -> float:
    """Wall time of the slowest rank (what bench.py reports)."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_results(local: list) -> list:
    """Concatenate per-rank result lists in rank order on every rank (the `cat` merge of run_qa_*.sh:41-48)."""
    rank, world = rank_world()
    if world == 1:
        return list(local)
    out = [None] * world
    dist.all_gather_object(out, list(local))
    return [x for part in out for x in part]


class FlatGradBucket:
    """One flat fp32 buffer for the gradients of the trainable set, all-reduced (sum) in a single collective.
    xGMI is point-to-point (7 links x ~153 GB/s per GPU): one large message per step, not one per tensor."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else "cpu"
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)

    def all_reduce(self, average: bool = True) -> None:
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is not None:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            else:
                self.flat[off:off + n].zero_()
            off += n
        rank, world = rank_world()
        if world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if average:
                self.flat.div_(world)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(self.flat[off:off + n].view_as(p))
            off += n
