"""Input contract of the path: frame picking and device-side preprocessing.

Host-side twins of ``sample_frames`` (src/data/components/util.py:20-34) and of the tail of ``get_frames``
(eval/utils/builder_utils.py:117-144).  Video decoding itself (PyAV) is not part of this package: ``get_frames``
starts from decoded RGB frames that are already in HBM and returns what the reference's ``get_frames`` returns.
"""
from typing import List, Tuple

import numpy as np
import torch
from torch import Tensor

from . import ops


def sample_frames(num_frames: int, video_len: int, sample: str = "rand", fix_start: float = -1) -> List[int]:
    """src/data/components/util.py:20-34.  ``fix_start >= 0`` wins over ``sample`` (as in the reference, whose eval
    caller passes ("uniform", 1.) and therefore gets ``intv[i] + 1``, not the midpoints)."""
    if num_frames >= video_len:
        return list(range(video_len))
    intv = np.linspace(start=0, stop=video_len, num=num_frames + 1).astype(int)
    if sample == "rand" and fix_start < 0:
        import random
        return [random.choice(range(intv[i], intv[i + 1])) for i in range(len(intv) - 1)]
    if fix_start >= 0:
        return [int(intv[i]) + int(fix_start) for i in range(len(intv) - 1)]
    if sample == "uniform":
        return [int((intv[i] + intv[i + 1] - 1) // 2) for i in range(len(intv) - 1)]
    raise NotImplementedError


def candidate_frame_ids(vlen: int, n_cand: int = 32) -> List[int]:
    """eval/utils/builder_utils.py:131-139: duplicate-double the index list until it has n_cand entries, then
    ``sample_frames(n_cand, len, "uniform", 1.)``."""
    indices = list(range(vlen))
    while len(indices) < n_cand:
        indices = [f for ind in indices for f in (ind, ind)]
    ids = sample_frames(n_cand, len(indices), "uniform", 1.0)
    return [indices[i] for i in ids]


@torch.no_grad()
def get_frames(raw: Tensor, target_size: int = 224, n_cand: int = 32) -> Tuple[Tensor, Tensor]:
    """Decoded frames ``raw`` [T, H0, W0, 3] uint8 on the device -> ``(frames [n_cand, 3, S, S], flow_frames [T, 3, S, S])``
    fp32, the return value of the reference's ``get_frames`` (eval/utils/builder_utils.py:117-144).  Both tensors come
    from the preprocessing kernel directly (the pick is applied while resizing, no gather pass)."""
    flow_frames = ops.preprocess_frames(raw, None, target_size)
    idx = torch.tensor(candidate_frame_ids(raw.shape[0], n_cand), dtype=torch.int64, device=raw.device)
    frames = ops.preprocess_frames(raw, idx, target_size)
    return frames, flow_frames


def pack_decoded(frames, out: np.ndarray = None) -> np.ndarray:
    """Any decoder's output -> one contiguous uint8 array [T, H, W, 3] (written into ``out[:T]`` if given): a [T, H, W, 3] array /
    CPU tensor, or an iterable of [H, W, 3] RGB frames (numpy arrays, CPU tensors, or objects with ``to_ndarray(format="rgb24")``
    such as PyAV's VideoFrame -- what eval/utils/builder_utils.py:117-128 collects)."""
    def one(f):
        if hasattr(f, "to_ndarray"):
            f = f.to_ndarray(format="rgb24")
        if isinstance(f, Tensor):
            f = f.numpy()
        f = np.asarray(f)
        if f.dtype != np.uint8 or f.ndim != 3 or f.shape[2] != 3:
            raise TypeError(f"decoded frames must be uint8 [H, W, 3] RGB (got {f.dtype} {f.shape})")
        return f
    if isinstance(frames, Tensor):
        frames = frames.numpy()
    if isinstance(frames, np.ndarray):
        if frames.dtype != np.uint8 or frames.ndim != 4 or frames.shape[3] != 3:
            raise TypeError(f"decoded clip must be uint8 [T, H, W, 3] (got {frames.dtype} {frames.shape})")
        if out is None:
            return np.ascontiguousarray(frames)
        if frames.shape[0] > out.shape[0] or frames.shape[1:] != out.shape[1:]:
            raise ValueError(f"clip {frames.shape} does not fit the staging slot {out.shape}")
        out[:frames.shape[0]] = frames
        return out[:frames.shape[0]]
    t = 0
    rows = []
    for f in frames:
        f = one(f)
        if out is None:
            rows.append(f)
        else:
            if t >= out.shape[0] or f.shape != out.shape[1:]:
                raise ValueError(f"frame {t} {f.shape} does not fit the staging slot {out.shape}")
            out[t] = f
        t += 1
    if t == 0:
        raise ValueError("empty clip")
    return np.stack(rows) if out is None else out[:t]


class FrameStager:
    """Decode feed of row f3 (eval/utils/builder_utils.py:117-144) for an EXTERNAL decoder: the decoder (PyAV, a hardware decoder's host
    copy, a dataloader worker) writes RGB frames into pinned host slots, a copy stream uploads a slot while the previous clip computes,
    and ``frames()`` hands the device-side preprocessing (resize / crop-free squash, /255, CLIP normalise, candidate pick: one kernel each
    for the 32 candidate frames and the T flow frames) a tensor that is already in HBM.

        st = FrameStager("cuda", max_frames=96, height=360, width=640)        # two slots by default
        t0 = st.stage(decoded_clip_0)                                          # async H2D; returns a ticket
        t1 = st.stage(decoded_clip_1)                                          # overlaps clip 0's compute
        frames, flow_frames = st.frames(t0)                                    # == builder_utils.get_frames(...) of the reference
    A slot is reused once the clip staged ``slots`` calls earlier has been consumed by ``frames()`` (stream-ordered: an event guards it); staging into a
    slot whose clip has not been consumed raises, and so does a ticket whose clip is gone."""

    def __init__(self, device="cuda", max_frames: int = 256, height: int = 360, width: int = 640, slots: int = 2):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("FrameStager stages into HBM: it needs a cuda device")
        self.shape = (max_frames, height, width, 3)
        self.host = [torch.empty(self.shape, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.dev = [torch.empty(self.shape, dtype=torch.uint8, device=self.device) for _ in range(slots)]
        self.uploaded = [torch.cuda.Event() for _ in range(slots)]
        self.consumed = [None] * slots
        self.state = ["free"] * slots          # free -> staged (stage) -> consumed (frames) -> staged ...
        self.gen = [0] * slots                 # generation of the clip a slot holds: a ticket of an overwritten clip is refused
        self.copy = torch.cuda.Stream(device=self.device)
        self.n = 0

    def stage(self, decoded) -> Tuple[int, int, int]:
        i = self.n % len(self.host)
        if self.state[i] == "staged":
            # (the slot's earlier clip has not been handed out by frames() yet -- overwriting the pinned buffer would race its upload and
            # silently give that ticket the newer clip's pixels)
            raise RuntimeError(f"FrameStager: slot {i} still holds a staged clip that frames() has not consumed ({len(self.host)} slots: call "
                               f"frames() for the oldest ticket first, or construct with more slots)")
        self.n += 1
        if self.consumed[i] is not None:
            self.consumed[i].synchronize()                      # the clip that used this slot has been preprocessed (host buffer AND device buffer free)
        t = pack_decoded(decoded, self.host[i].numpy()).shape[0]
        with torch.cuda.stream(self.copy):
            self.dev[i][:t].copy_(self.host[i][:t], non_blocking=True)
            self.uploaded[i].record(self.copy)
        self.state[i] = "staged"
        self.gen[i] += 1
        return i, t, self.gen[i]

    @torch.no_grad()
    def frames(self, ticket, target_size: int = 224, n_cand: int = 32) -> Tuple[Tensor, Tensor]:
        i, t = ticket[0], ticket[1]
        if len(ticket) > 2 and (ticket[2] != self.gen[i] or self.state[i] != "staged"):
            raise RuntimeError(f"FrameStager: stale ticket for slot {i} (its clip has been consumed or overwritten)")
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self.uploaded[i])
        out = get_frames(self.dev[i][:t], target_size, n_cand)
        ev = torch.cuda.Event()
        ev.record(cur)
        self.consumed[i] = ev
        self.state[i] = "consumed"
        return out
