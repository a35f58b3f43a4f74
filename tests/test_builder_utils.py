"""CPU: host logic of the eval-driver twins (videotgb_amd/builder_utils.py) that needs no GPU."""
import pytest
import torch


class Tok:
    bos_token_id = 1

    def __call__(self, text):
        ids = [1] + [ord(c) for c in text]
        return type("E", (), {"input_ids": ids})()

    def batch_decode(self, ids, skip_special_tokens=True):
        return ["".join(chr(t) for t in row.tolist() if t > 2) for row in ids]


def test_keywords_stopping_criteria_matches_reference_semantics():
    """eval/utils/builder_utils.py:320-346: BOS stripped from the keyword ids, exact-suffix match on ids, substring match on
    the decoded tail, batch size 1 only."""
    from videotgb_amd.builder_utils import KeywordsStoppingCriteria
    tok = Tok()
    prompt = torch.tensor([[1, 50, 51]])
    crit = KeywordsStoppingCriteria(["</s>"], tok, prompt)
    assert crit.keyword_ids[0].tolist() == [ord(c) for c in "</s>"] and crit.start_len == 3
    out = torch.cat([prompt, torch.tensor([[ord("o"), ord("k")]])], 1)
    assert crit(out, None) is False
    out = torch.cat([out, torch.tensor([[ord(c) for c in "</s>"]])], 1)
    assert crit(out, None) is True
    with pytest.raises(AssertionError):
        crit(torch.cat([out, out], 0), None)


def test_load_pretrained_model_rejects_unknown_base_and_get_frames_needs_pyav(tmp_path):
    from videotgb_amd import builder_utils
    with pytest.raises(ValueError, match="neither an instructblip nor a blip2"):
        builder_utils.load_pretrained_model(str(tmp_path / "x.ckpt"), str(tmp_path / "llava"), None, "cpu", load_processors=False)
    try:
        import av  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="PyAV"):
            builder_utils.get_frames(str(tmp_path / "clip.mp4"), fps=2)


def test_module_flavour_table_matches_the_reference_variants():
    """which LightningModule runs which flavour of the path (flow source, TGB mode, V, index map, pooling)."""
    from videotgb_amd import modules as M
    f = {k: (c.ARCH, c.SAMPLER, c.EVAL_RAFT, c.TGB_MODE if c.SAMPLER else None, c.MAP if c.SAMPLER else None, c.V_FROM_LENGTHS, c.WIDTHS, c.LORA)
         for k, c in M.TARGETS.items()}
    assert f["src.models.LSTP_module.LSTPModule"] == ("instructblip", True, True, "multi_modal", "A", False, False, None)
    assert f["src.models.LSTP_blip2_module.LSTPModule"] == ("blip2", False, False, None, None, False, False, None)
    assert f["src.models.LSTP_SF_module.LSTPSFModule"] == ("instructblip", True, False, "fusion", "B", True, False, None)
    assert f["src.models.LSTP_SF_blip2_module.LSTPSFModule"] == ("blip2", True, False, "fusion", "B", True, False, None)
    assert f["src.models.LSTP_Vicuna_IVT_module.LSTPModule"][6:] == (True, "CAUSAL_LM")
    assert f["src.models.LSTP_Blip2_IVT_module.LSTPModule"][6:] == (True, "SEQ_2_SEQ_LM")
    assert f["src.models.LSTP_Vicuna_IV_module.LSTPModule"][6:] == (True, None)


def test_pack_decoded_accepts_any_decoder_output():
    """row f3's decode feed: arrays, CPU tensors, iterables of frames and PyAV-like frame objects all become one uint8 [T, H, W, 3] clip."""
    import numpy as np
    import pytest
    import torch
    from videotgb_amd import video
    rng = np.random.default_rng(0)
    clip = rng.integers(0, 256, (5, 12, 16, 3), dtype=np.uint8)

    class AvFrame:                                                 # what av.VideoFrame offers
        def __init__(self, a):
            self.a = a

        def to_ndarray(self, format="rgb24"):
            assert format == "rgb24"
            return self.a
    for src in (clip, torch.from_numpy(clip), list(clip), (torch.from_numpy(f) for f in clip), [AvFrame(f) for f in clip]):
        assert np.array_equal(video.pack_decoded(src), clip)
    slot = np.zeros((8, 12, 16, 3), dtype=np.uint8)
    out = video.pack_decoded(iter(clip), slot)
    assert out.shape == clip.shape and np.array_equal(slot[:5], clip) and out.base is slot or np.shares_memory(out, slot)
    with pytest.raises(ValueError):
        video.pack_decoded(clip, np.zeros((4, 12, 16, 3), dtype=np.uint8))           # more frames than the slot holds
    with pytest.raises(TypeError):
        video.pack_decoded(clip.astype(np.float32))
    with pytest.raises(ValueError):
        video.pack_decoded([])
