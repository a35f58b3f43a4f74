"""Multi-GPU side of the path: one process per GPU, clips sharded with no data-path collective.

Inference: the clip list is cut into contiguous chunks of ceil(len / n) clips, chunk k
goes to rank k (eval/inference.py:21-29, driven per GPU by eval/scripts/run_qa_*.sh:16-48); ranks never
exchange data, results are concatenated in rank order.  Training (config 5): the only exchange is
the sum all-reduce of the trainable gradients (785 MB with the reference's trainable set, 16.8 MB LoRA-only) once per
optimizer step -- gradients live in one flat fp32 buffer and are reduced in a few large segments under the remaining
backward: RCCL through the C ABI (vtgb_allreduce_f32, ``RcclComm``) on the MI355X node, torch.distributed ("gloo") in the
CPU tests.
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


class RcclComm:
    """One RCCL communicator per process through the C ABI (vtgb_comm_*, include/vtgb.h): rank 0 creates the unique id, the
    128 bytes travel over the already initialised torch.distributed group (any backend: it is host data), every rank joins.
    ``all_reduce_(t, average, stream)`` reduces a contiguous fp32 CUDA tensor in place, asynchronously on ``stream``."""

    def __init__(self, device=None):
        import ctypes as C
        from . import _lib as L
        rank, world = rank_world()
        self.rank, self.world = rank, world
        if device is not None:
            torch.cuda.set_device(device)
        ident = (C.c_ubyte * L.COMM_ID_BYTES)()
        if rank == 0:
            L.check(L.lib().vtgb_comm_unique_id(ident))
        if world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0)
            ident = (C.c_ubyte * L.COMM_ID_BYTES).from_buffer_copy(box[0])
        self._h = C.c_void_p()
        L.check(L.lib().vtgb_comm_init(C.byref(self._h), ident, rank, world))

    def all_reduce_(self, t: torch.Tensor, average: bool = True, stream=None) -> None:
        import ctypes as C
        from . import _lib as L
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("RcclComm.all_reduce_: a contiguous fp32 device tensor is required")
        st = stream if stream is not None else torch.cuda.current_stream(t.device)
        L.check(L.lib().vtgb_allreduce_f32(self._h, C.c_void_p(t.data_ptr()), t.numel(), 1 if average else 0, C.c_void_p(st.cuda_stream)))

    def close(self) -> None:
        from . import _lib as L
        if self._h:
            L.check(L.lib().vtgb_comm_destroy(self._h))
            self._h = None


class FlatGradBucket:
    """The gradients of the trainable set as VIEWS of one flat fp32 buffer, reduced over the ranks in a few large segments
    that start as soon as their gradients are complete -- what DDP's bucketed, overlapped all-reduce does for the reference
    (configs/trainer/ddp.yaml:4), sized for xGMI: point-to-point links, ring collectives are per-link bound, so segments are
    tens of MB (``segment_bytes``), not DDP's 25 MB default times dozens.

    * ``p.grad`` of every parameter is a view into ``flat`` (autograd accumulates in place): no copy-in / copy-out around the
      collective (round 2 copied the 785 MB bucket twice per step).  Zero gradients with ``zero_()`` or
      ``optimizer.zero_grad(set_to_none=False)``; a ``p.grad`` that was replaced behind the bucket's back is copied in and
      re-attached by ``all_reduce``.
    * Segments are cut in REVERSE parameter order (backward reaches the last parameters first).  ``arm()`` before the backward
      whose gradients are to be exchanged (the last micro-batch of an accumulation window): post-accumulate-grad hooks count a
      segment's parameters and launch its all-reduce on a side stream (RCCL through the C ABI when ``comm`` is given, else
      ``torch.distributed`` with ``async_op``) under the remaining backward.  ``all_reduce()`` launches what is left (segments
      holding parameters that received no gradient) and waits.
    * Collectives are matched across ranks by ISSUE ORDER, so segments are always launched in index order: segment s goes out
      only once segments 0..s-1 have (a later segment that completes first stays pending), and ``all_reduce()`` flushes the
      rest in index order.  A rank on which some parameter got no gradient this step (the Q-Former when every clip of the rank
      has width 0) therefore issues the same sequence as its peers, just later.  Readiness is a SET of parameter ids per
      segment: a hook that fires twice (a second backward, a tied weight under re-entrant checkpointing) cannot launch early."""

    def __init__(self, params: Iterable[torch.nn.Parameter], segment_bytes: int = 64 << 20, comm: "RcclComm" = None):
        self.params = [p for p in params if p.requires_grad]
        bad = [tuple(p.shape) for p in self.params if p.dtype != torch.float32]
        if bad:      # p.grad = <fp32 view> would raise "assigned grad has data of a different type" deep inside _attach
            raise TypeError(f"FlatGradBucket: trainable parameters must be fp32 (the bucket is one flat fp32 buffer); got {len(bad)} "
                            f"non-fp32 parameter(s), first shape {bad[0]} -- keep LoRA / prefix stages in fp32 or cast their master copy")
        sizes = [p.numel() for p in self.params]
        n = sum(sizes)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.comm = comm
        self.offsets, off = [], 0
        for s_ in sizes:
            self.offsets.append(off)
            off += s_
        self._attach()
        # segments: contiguous parameter runs, cut from the END of the list
        per = max(1, segment_bytes // 4)
        self.segments, hi = [], len(self.params)
        while hi > 0:
            lo, acc = hi, 0
            while lo > 0 and (acc == 0 or acc + sizes[lo - 1] <= per):
                lo -= 1
                acc += sizes[lo]
            self.segments.append((lo, hi))
            hi = lo
        self._seg_of = {}
        for si, (lo, hi) in enumerate(self.segments):
            for k in range(lo, hi):
                self._seg_of[k] = si
        self._index = {id(p): k for k, p in enumerate(self.params)}
        self._armed = False
        self._ready = [set() for _ in self.segments]
        self._launched = [False] * len(self.segments)
        self._next = 0                                   # segments [0, _next) have been launched
        self._works = []
        self._side = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._average = True
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._on_grad)

    # ---- gradient views
    def _view(self, k: int) -> torch.Tensor:
        p = self.params[k]
        return self.flat[self.offsets[k]:self.offsets[k] + p.numel()].view_as(p)

    def _attach(self) -> None:
        for k, p in enumerate(self.params):
            v = self._view(k)
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
            p.grad = v

    def zero_(self) -> None:
        self.flat.zero_()

    # ---- overlapped exchange
    def arm(self, average: bool = True) -> None:
        """The next backward's gradients are final: start each segment's all-reduce as soon as its parameters have theirs."""
        self._armed, self._average = True, average
        self._ready = [set() for _ in self.segments]
        self._launched = [False] * len(self.segments)
        self._next = 0

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        if not self._armed:
            return
        k = self._index.get(id(p))
        if k is None:
            return
        if p.grad.data_ptr() != self.flat.data_ptr() + 4 * self.offsets[k]:      # replaced behind our back: fold it in, re-attach
            v = self._view(k)
            v.copy_(p.grad)
            p.grad = v
        si = self._seg_of[k]
        self._ready[si].add(k)
        self._launch_ready()

    def _complete(self, si: int) -> bool:
        lo, hi = self.segments[si]
        return len(self._ready[si]) == hi - lo

    def _launch_ready(self) -> None:
        """Launch, in index order, every complete segment whose predecessors have all been launched."""
        while self._next < len(self.segments) and self._complete(self._next):
            self._launch(self._next)
            self._next += 1

    def _launch(self, si: int) -> None:
        self._launched[si] = True
        rank, world = rank_world()
        if world <= 1 and self.comm is None:
            return
        lo, hi = self.segments[si]
        seg = self.flat[self.offsets[lo]:self.offsets[hi - 1] + self.params[hi - 1].numel()]
        if self.comm is not None:
            cur = torch.cuda.current_stream(self.flat.device)
            self._side.wait_stream(cur)                       # the segment's gradients are complete on the compute stream
            self.comm.all_reduce_(seg, self._average, self._side)
            seg.record_stream(self._side)
        else:
            self._works.append((dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True), seg))

    def all_reduce(self, average: bool = True) -> None:
        """Finish the exchange: attach stray gradients, launch the segments the hooks did not (no ``arm()``, or parameters that
        received no gradient this step), wait for all of them; ``flat`` then holds the sum (mean) over the ranks."""
        if not self._armed:
            self.arm(average)
        elif average != self._average:
            raise ValueError(f"FlatGradBucket: arm(average={self._average}) and all_reduce(average={average}) disagree "
                             "(segments already in flight were launched with the armed value)")
        self._attach()
        for si in range(self._next, len(self.segments)):      # what the hooks did not launch, in index order
            self._launch(si)
        self._next = len(self.segments)
        rank, world = rank_world()
        if self.comm is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self._side)
        else:
            for w, seg in self._works:
                w.wait()
                if self._average and world > 1:
                    seg.div_(world)
            self._works = []
        self._armed = False
