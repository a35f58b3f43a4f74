"""Self-refinement pseudo-labels: twins of ``src/models/LSTP_SF_module.py:147-298``.

The reference asks the frozen Q-Former + LLM what it would answer from EACH of the 32 candidate frames alone (a Python
double loop of ``generate`` calls, nframe sequences at a time, its own comment: "TODO: use loop to prevent from OOM"),
scores every answer against the ground truth with ``rouge_n`` (src/gadgets/my_metrics.py:131-157), turns the 32 scores
into a pseudo span with a monotone stack (largest-rectangle-in-histogram, :246-261), rescales it to flow positions
(:263-265) and trains the TGB's start/end logits against it (:285-298).

Here the B*32 frames go through the HIP ViT / Q-Former / projection in one batch and through one hipGraph-replayed
greedy decode; scoring and span extraction are host logic on 32 numbers per clip (string work needs the tokenizer).
"""
from typing import Callable, List, Optional, Sequence, Tuple

import torch
from torch import Tensor


def _recall(gold_tokens: Sequence[str], pred_tokens: Sequence[str], ignore) -> float:
    """Unigram recall of one (gold, prediction) pair: the fraction of gold tokens (those not in ``ignore``) that occur
    anywhere in the prediction; repeated gold tokens count each time.  No scored token -> 0."""
    scored = [t for t in gold_tokens if ignore is None or t not in ignore]
    if ignore is None:
        return sum(t in pred_tokens for t in scored) / len(gold_tokens)          # (an empty gold string divides by zero, as in the reference)
    return sum(t in pred_tokens for t in scored) / len(scored) if scored else 0


def rouge_n(gold, pred, ignore=(",", ".")):
    """The pseudo-label score of the self-refinement loop, with the observable behaviour of src/gadgets/my_metrics.py:131-185
    (pseudo-label parity needs its quirks, kept here as NAMED branches rather than transcribed):
      * strings: plain unigram recall (``_recall``);
      * lists, ``ignore`` given -- quirk LIST_NORMALISATION: each pair's recall is additionally divided by the NUMBER OF PAIRS
        (my_metrics.py:154-155), so list scores shrink with the batch size;
      * lists, ``ignore=None`` -- quirk FIRST_PAIR_ONLY: the reference returns from inside its loop, i.e. a single float, the
        recall of the first pair (an empty list falls through to ``[]``)."""
    if not isinstance(gold, list):
        return _recall(gold.split(), pred.split(), ignore)
    pairs = [(g.split(), p.split()) for g, p in zip(gold, pred)]
    if ignore is None:                                  # FIRST_PAIR_ONLY
        return _recall(*pairs[0], None) if pairs else []
    n_pairs = len(gold)                                 # LIST_NORMALISATION
    return [_recall(g, p, ignore) / n_pairs for g, p in pairs]


def monotone_span(score: Sequence[float]) -> Tuple[int, int]:
    """LSTP_SF_module.py:249-261: the window maximising width * min(score) (monotone stack over the scores padded
    with a 0 on both sides); ties keep the first window found; indices refer to the unpadded sequence."""
    bs, start_target, end_target = 0, 0, len(score) - 1
    stack: List[int] = []
    score = [0] + list(score) + [0]
    for i in range(len(score)):
        while stack and score[stack[-1]] > score[i]:
            tmp = stack.pop()
            tmp_bs = (i - stack[-1] - 1) * score[tmp]
            if tmp_bs > bs:
                bs = tmp_bs
                start_target, end_target = stack[-1], i - 2
        stack.append(i)
    return start_target, end_target


def pseudo_spans(scores: Tensor, flow_lengths: Sequence[int], device=None) -> Tuple[Tensor, Tensor]:
    """scores [B, num_frames] (float32, as ``torch.tensor(scores, dtype=torch.float)`` :243-244) -> start / end targets in
    flow coordinates, ``int(t / (num_frames - 1) * (flow_length - 1))`` (:263-268)."""
    b, n = scores.shape
    st, en = zip(*(monotone_span(row.tolist()) for row in scores.float().cpu()))
    st = [int(st[i] / (n - 1) * (flow_lengths[i] - 1)) for i in range(b)]
    en = [int(en[i] / (n - 1) * (flow_lengths[i] - 1)) for i in range(b)]
    return torch.tensor(st, dtype=torch.long, device=device), torch.tensor(en, dtype=torch.long, device=device)


def mrc_loss(of_logits: Tensor, start_targets: Tensor, end_targets: Tensor) -> Tensor:
    """LSTP_SF_module.py:285-298: CE over the L start logits and the L end logits, targets clamped to [0, L],
    ignore_index = L, mean of the two."""
    start_logits, end_logits = of_logits.split(1, dim=-1)
    ignored_index = start_logits.size(1)
    loss_fct = torch.nn.CrossEntropyLoss(ignore_index=ignored_index)
    if start_targets.dim() > 1:
        start_targets = start_targets.squeeze(-1)
    if end_targets.dim() > 1:
        end_targets = end_targets.squeeze(-1)
    start_targets = start_targets.clamp(0, ignored_index)
    end_targets = end_targets.clamp(0, ignored_index)
    start_loss = loss_fct(start_logits.squeeze(-1).contiguous(), start_targets)
    end_loss = loss_fct(end_logits.squeeze(-1).contiguous(), end_targets)
    return (start_loss + end_loss) / 2


@torch.no_grad()
def frame_answers(lstp, frames: Tensor, batch_size: int, qformer_text: Optional[Tensor], qformer_text_mask: Optional[Tensor],
                  question: Tensor, question_mask: Tensor, max_length: int = 128) -> Tensor:
    """Greedy answer tokens from every candidate frame on its own (LSTP_SF_module.py:150-204), all B*num_frames frames in
    one batch: ViT -> Q-Former (the clip's instruction repeated per frame) -> language_projection (a "clip" of one frame)
    -> [prefix | question] -> greedy decode.  ``max_length`` counts prefix + question + new tokens, as
    ``generate(inputs_embeds=..., max_length=128)`` does in the pinned transformers.  Requires unpadded questions
    (the graph decoder has no padding mask); returns ids [B*num_frames, <= n_new] (rows end at EOS and are padded, as HF
    generate returns them) with the LLaMA 0 -> 2 patch applied."""
    from .decode import GreedyDecoder
    n_all = frames.shape[0]
    num_frames = n_all // batch_size
    if not bool((question_mask != 0).all()):
        raise NotImplementedError("frame_answers: padded questions need HF generate (attention-mask aware)")
    enc = None
    if qformer_text is not None:
        enc = {"qformer_input_ids": torch.repeat_interleave(qformer_text, num_frames, 0),
               "qformer_attention_mask": torch.repeat_interleave(qformer_text_mask, num_frames, 0)}
    lm_inputs = lstp.prefix(frames, n_all, 1, enc, "mean")                 # nframe = 1: every frame is its own prefix
    lm = lstp.model.language_model
    dt = next(lm.parameters()).dtype
    q = torch.repeat_interleave(question, num_frames, 0)
    emb = torch.cat([lm_inputs.to(dt), lstp.model.get_input_embeddings()(q).to(dt)], dim=1)
    n_new = max_length - emb.shape[1]
    if n_new <= 0:
        raise ValueError(f"max_length={max_length} leaves no room after the {emb.shape[1]}-token prompt")
    if getattr(lstp, "_decoder", None) is None or lstp._decoder.lm is not lm:
        lstp._decoder = GreedyDecoder(lm)
    gc = getattr(lm, "generation_config", None)     # HF generate's defaults: stop at EOS, pad afterwards
    out = lstp._decoder.generate(emb, n_new, eos_token_id=getattr(gc, "eos_token_id", None), pad_token_id=getattr(gc, "pad_token_id", None) or 0)
    if lstp.model.config.text_config.architectures[0] == "LLaMAForCausalLM":
        out[out == 0] = 2
    return out


def pseudo_labels(predict: List[str], text_answer: List[str], batch_size: int, num_frames: int, flow_lengths: Sequence[int],
                  device=None) -> Tuple[Tensor, Tensor, Tensor]:
    """LSTP_SF_module.py:239-268 from decoded strings: (scores [B, num_frames], start_targets, end_targets)."""
    target = [text_answer[int(idx // num_frames)] for idx in range(len(predict))]
    scores = torch.tensor(rouge_n(target, predict), dtype=torch.float).view(batch_size, num_frames)
    st, en = pseudo_spans(scores, flow_lengths, device)
    return scores, st, en


def self_refine_targets(lstp, batch: dict, batch_decode: Callable[[Tensor], List[str]], num_frames: int = 32, max_length: int = 128):
    """Step 1 of LSTPSFModule.forward (:147-268) for one batch dict (keys as produced by the reference's collate)."""
    b = batch["question"].shape[0]
    ids = frame_answers(lstp, batch["frames"], b, batch.get("qformer_text"), batch.get("qformer_text_attention_mask"),
                        batch["question"], batch["question_attention_mask"], max_length)
    predict = batch_decode(ids)
    return pseudo_labels(predict, batch["text_answer"], b, num_frames, batch["of_lengths"], batch["frames"].device)
