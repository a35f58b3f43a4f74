"""f4 (self-refinement pseudo labels), CPU side: oracle and product host logic vs vectors produced by the reference's
rouge_n and by executing the monotone-stack / rescale / MRC-loss lines of LSTPSFModule.forward verbatim."""
import os

import torch

from conftest import GOLDEN, load_golden
from oracle import vtgb_oracle as O
from videotgb_amd import refine


def texts():
    lines = open(os.path.join(GOLDEN, "refine_text.txt")).read().split("\n")[:-1]
    return lines[:4], lines[4:]


def test_rouge_spans_and_mrc_loss_match_reference():
    g = load_golden("refine")
    gold, pred = texts()
    B, N = 4, 32
    target = [gold[i // N] for i in range(len(pred))]
    want = g["rouge"].tolist()
    assert O.rouge_n_list(target, pred) == want                      # same divisions in the same order: exact
    assert refine.rouge_n(target, pred) == want
    assert refine.rouge_n(target[0], pred[0]) == want[0] * len(pred)   # the scalar form lacks the list form's / len(gold)
    fl = g["flow_lengths"].tolist()
    scores = torch.tensor(want, dtype=torch.float).view(B, N)
    assert O.pseudo_span_targets(scores, fl) == (g["start_targets"].tolist(), g["end_targets"].tolist())
    sc, st, en = refine.pseudo_labels(pred, gold, B, N, fl)
    assert torch.equal(sc, scores) and st.tolist() == g["start_targets"].tolist() and en.tolist() == g["end_targets"].tolist()
    rows, rl = g["rows"], g["rows_lengths"].tolist()
    assert O.pseudo_span_targets(rows, rl) == (g["rows_start"].tolist(), g["rows_end"].tolist())
    st, en = refine.pseudo_spans(rows, rl)
    assert st.tolist() == g["rows_start"].tolist() and en.tolist() == g["rows_end"].tolist()
    assert refine.monotone_span([0.0] * 32) == (0, 31)               # nothing scored: the full span stays
    for fn in (O.mrc_loss, refine.mrc_loss):
        loss = fn(g["of_logits"], g["mrc_start"].clone(), g["mrc_end"].clone())
        assert torch.equal(loss.reshape(1), g["mrc_loss"])
