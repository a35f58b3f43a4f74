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
