"""Twins of what the reference's eval driver imports (eval/inference.py:16):
``from .utils.builder_utils import load_pretrained_model, get_frames, KeywordsStoppingCriteria``.

* ``load_pretrained_model`` (eval/utils/builder_utils.py:169-187): same arguments, same return triple, the same strict load
  of the Lightning checkpoint's ``state_dict`` -- into the HIP-backed ``LSTP`` / ``LSTP_blip2`` of videotgb_amd.models.
* ``get_frames`` (:117-144): the reference decodes the file with PyAV and transforms on the CPU; here the transform chain
  and the 32-frame pick run on the device (videotgb_amd.video / vtgb_preprocess_frames).  A path is decoded with PyAV when
  it is installed (it is not part of this package); a uint8 tensor [T, H, W, 3] of decoded frames is taken as is.
* ``KeywordsStoppingCriteria`` (:320-346): the same contract, written for this package (token test, then text test).
"""
from __future__ import annotations

import torch

from . import models, video


def load_pretrained_model(ckpt_path, base_model_path, base_sampler_path, device, lora=False, compute_dtype="bf16", load_processors=True,
                          **model_kwargs):
    """eval/utils/builder_utils.py:169-187.  ``compute_dtype`` / ``load_processors`` / ``model_kwargs`` are keyword extensions
    (the processors need tokenizer files next to the config; tests that only have a config.json pass ``load_processors=False``)."""
    print("start to load model...")
    processor = sampler_processor = None
    if load_processors:
        from transformers import AutoProcessor, AutoTokenizer
        processor = AutoProcessor.from_pretrained(base_model_path)
        sampler_processor = AutoTokenizer.from_pretrained(base_sampler_path)
    if "instructblip" in base_model_path:
        model = models.LSTP(base_model_path, device, lora, compute_dtype=compute_dtype, **model_kwargs)
    elif "blip2" in base_model_path:
        model = models.LSTP_blip2(base_model_path, device, lora, compute_dtype=compute_dtype, **model_kwargs)
    else:   # the reference leaves `model` unbound here and dies with UnboundLocalError
        raise ValueError(f"base_model_path {base_model_path!r} names neither an instructblip nor a blip2 model")
    state_dict = torch.load(ckpt_path, map_location="cpu")
    msg = model.load_state_dict(state_dict["state_dict"])
    print(">>> Load checkpoint for LSTP from", ckpt_path)
    miss = set(m.split(".")[0] for m in msg.missing_keys)
    unexp = set(m.split(".")[0] for m in msg.unexpected_keys)
    print("Missing:", miss if len(miss) else "None")
    print("Unexpected:", unexp if len(unexp) else "None")
    return model, processor, sampler_processor


def read_video_frames(video_path: str, fps=2) -> torch.Tensor:
    """read_videos_av (eval/utils/builder_utils.py:68-87): decode with PyAV; when ``fps`` does not exceed the stream's average
    rate every ``int(average_rate)``-th frame is kept (one frame per second, whatever ``fps`` says -- as the reference does),
    otherwise every frame.  Returns uint8 [T, H, W, 3].  PyAV is outside this package."""
    try:
        import av
    except ImportError as e:
        raise ImportError("decoding a video file needs PyAV (`av`); pass decoded frames [T, H, W, 3] uint8 instead") from e
    import numpy as np
    with av.open(video_path) as container:
        avg_fps = int(container.streams.video[0].average_rate)
        step = avg_fps if (fps is not None and fps <= avg_fps) else 1
        frames = [f.to_ndarray(format="rgb24") for i, f in enumerate(container.decode(video=0)) if i % max(step, 1) == 0]
    return torch.from_numpy(np.stack(frames, axis=0))


def get_frames(video_path, target_size=224, keyframe=False, start_ratio=0.0, end_ratio=1.0, fps=None, device="cuda"):
    """eval/utils/builder_utils.py:117-144 -> ``(frames [32, 3, S, S], flow_frames [T, 3, S, S])`` fp32, here on ``device``
    (the reference returns CPU tensors and the driver moves them, eval/inference.py:70-71).  ``video_path``: a file path (decoded with
    PyAV, as the reference does -- if it is installed), or the output of ANY decoder: a [T, H, W, 3] uint8 tensor / numpy array, or an
    iterable of [H, W, 3] RGB frames (``video.pack_decoded``).  Host-side clips go through pinned memory (``video.FrameStager`` keeps
    the slots and overlaps the upload with the previous clip).  ``keyframe`` / ``start_ratio`` / ``end_ratio`` are accepted and, as in
    the reference's live code, unused."""
    raw = read_video_frames(video_path, fps) if isinstance(video_path, str) else video_path
    if not (isinstance(raw, torch.Tensor) and raw.is_cuda):
        raw = torch.from_numpy(video.pack_decoded(raw))
        if torch.device(device).type == "cuda":
            raw = raw.pin_memory()
    return video.get_frames(raw.to(device, non_blocking=True), target_size)


class KeywordsStoppingCriteria:
    """Stop generation when the answer ends with, or its recent text contains, one of ``keywords`` -- the contract of the class
    the eval driver imports under this name (eval/utils/builder_utils.py:320-346; batch size 1, callable as
    ``criteria(output_ids, scores) -> bool``).  Two tests per call, cheapest first:
      1. token test: the last ``len(ids_k)`` generated ids equal keyword k's token ids (a leading BOS id is not part of a keyword);
      2. text test: the decoded window of the last ``min(n_generated, longest keyword)`` ids contains a keyword as a substring."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = list(keywords)
        self.tokenizer = tokenizer
        self.start_len = int(input_ids.shape[1])
        self._id_tails = [self._keyword_ids(tokenizer, kw) for kw in self.keywords]
        self.max_keyword_len = max((t.numel() for t in self._id_tails), default=0)

    @staticmethod
    def _keyword_ids(tokenizer, keyword) -> torch.Tensor:
        ids = list(tokenizer(keyword).input_ids)
        if len(ids) > 1 and ids[0] == tokenizer.bos_token_id:
            ids = ids[1:]
        return torch.tensor(ids, dtype=torch.long)

    @property
    def keyword_ids(self):                                     # the reference's attribute name
        return self._id_tails

    def _token_hit(self, row: torch.Tensor) -> bool:
        for n, tail in enumerate(self._id_tails):
            if tail.device != row.device:
                self._id_tails[n] = tail = tail.to(row.device)
            if torch.equal(row[-tail.numel():], tail):
                return True
        return False

    def _text_hit(self, output_ids: torch.Tensor) -> bool:
        window = min(output_ids.shape[1] - self.start_len, self.max_keyword_len)
        text = self.tokenizer.batch_decode(output_ids[:, -window:], skip_special_tokens=True)[0]    # window 0 -> the whole row, as `[-0:]` is
        return any(kw in text for kw in self.keywords)

    def __call__(self, output_ids: torch.LongTensor, scores: torch.FloatTensor = None, **kwargs) -> bool:
        if output_ids.shape[0] != 1:
            raise AssertionError("Only support batch size 1 (yet)")
        return self._token_hit(output_ids[0]) or self._text_hit(output_ids)
