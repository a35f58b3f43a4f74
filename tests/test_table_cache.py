"""The packed-weight table of a stage is a cache of its parameters: anything that changes them must invalidate it (round-3 ADVICE):
an optimizer step, a PARENT's load_state_dict (nn.Module calls the child's _load_from_state_dict, not its load_state_dict),
`param.data = ...` -- for frozen stages too."""
import torch

from videotgb_amd import models, synth


def _stage():
    cfg = synth.tiny_cfg("instructblip")
    return models.VisionModel(cfg.vit, "bf16")


def test_parent_load_state_dict_invalidates_a_frozen_stage_table():
    st = _stage()
    for p in st.parameters():
        p.requires_grad_(False)
    parent = torch.nn.Module()
    parent.add_module("vision_model", st)
    st._table = object()                      # stands for the packed table built by a forward
    assert st._table is not None
    ckpt = {k: torch.randn_like(v) if v.is_floating_point() else v.clone() for k, v in parent.state_dict().items()}
    parent.load_state_dict(ckpt)
    assert st._table is None


def test_data_swap_and_inplace_update_invalidate_the_table():
    st = _stage()
    p = next(st.parameters())
    st._table = object()
    p.data = torch.randn_like(p)              # new storage, version counter unchanged
    assert st._table is None
    st._table = object()
    with torch.no_grad():
        p.add_(1.0)                           # what an optimizer step does
    assert st._table is None
    st._table = object()
    assert st._table is not None              # nothing changed: the cache stays
