"""CPU: the drop-in registration (videotgb_amd/dropin.py).  The first test needs nothing but this repo; the second runs the
REFERENCE's own ``eval/utils/builder_utils.load_pretrained_model`` -- unmodified, imported from /root/reference -- and
checks that it ends up constructing and loading the HIP-backed ``LSTP`` (skipped where the reference checkout is absent,
e.g. on the GPU box: nothing under /root/reference travels)."""
import importlib
import importlib.machinery
import os
import sys
import types

import pytest
import torch

from conftest import full_state_dict, write_hf_config

REF = "/root/reference"


def test_install_registers_the_reference_module_paths():
    from videotgb_amd import dropin, models, modules
    names = dropin.install()
    try:
        assert "eval.utils.model" in names and len(names) == 1 + len({t.rsplit(".", 1)[0] for t in modules.TARGETS})
        for target, cls in modules.TARGETS.items():      # what hydra.utils.instantiate(_target_=...) does: import module, getattr
            mod, attr = target.rsplit(".", 1)
            assert getattr(importlib.import_module(mod), attr) is cls
        from eval.utils.model import LSTP, LSTP_blip2     # noqa: the statement eval/utils/builder_utils.py:14 executes
        assert LSTP is models.LSTP and LSTP_blip2 is models.LSTP_blip2
    finally:
        dropin.uninstall()
    assert "eval.utils.model" not in sys.modules and "src.models.LSTP_module" not in sys.modules
    # constructor signatures are the reference's (eval/utils/model.py:21-26; src/models/LSTP_module.py:85-95)
    import inspect
    assert list(inspect.signature(models.LSTP.__init__).parameters)[1:4] == ["base_model_path", "device", "lora"]
    assert list(inspect.signature(modules.LSTPModule.__init__).parameters)[1:9] == [
        "model_name_or_path", "sampler_name_or_path", "of_extractor_name_or_path", "temperature", "optimizer", "scheduler",
        "scheduler_params", "generate_configs"]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "eval", "utils")), reason="reference checkout not present")
def test_reference_builder_utils_builds_the_hip_model_unchanged(tmp_path):
    from videotgb_amd import dropin, models
    from videotgb_amd.synth import tiny_cfg
    import transformers
    _ = (transformers.AutoProcessor, transformers.AutoTokenizer, transformers.StoppingCriteria)   # resolve before torchvision is stubbed
    stubbed = []
    for name in ("av", "decord", "cv2", "ffmpeg", "sentence_transformers", "peft", "torchvision", "torchvision.transforms"):
        if name not in sys.modules:                         # heavy optional imports of eval/utils/builder_utils.py, unused here
            m = types.ModuleType(name)
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)
            sys.modules[name] = m
            stubbed.append(name)
    sys.modules["decord"].cpu = lambda *a, **k: None
    sys.modules["decord"].bridge = types.SimpleNamespace(set_bridge=lambda *a, **k: None)
    for n in ("Compose", "RandomCrop", "RandomResizedCrop", "Normalize"):
        setattr(sys.modules["torchvision.transforms"], n, type(n, (), {}))
    for n in ("PeftModel", "PeftConfig", "get_peft_model", "get_peft_model_state_dict", "LoraConfig", "TaskType"):
        setattr(sys.modules["peft"], n, object)
    sys.path.insert(0, REF)
    dropin.install()
    try:
        try:
            bu = importlib.import_module("eval.utils.builder_utils")          # the reference's file, as is
        except Exception as e:   # an import of the reference's own environment that this image lacks
            pytest.skip(f"reference builder_utils not importable here: {e!r}")
        assert bu.LSTP is models.LSTP and bu.LSTP_blip2 is models.LSTP_blip2
        cfg = tiny_cfg("instructblip")
        cfg.vit.image = 56
        base = write_hf_config(str(tmp_path / "instructblip-tiny"), "instructblip", cfg)
        # the reference's constructor call has no way to shrink the TGB: it is BERT-base (BertConfig(fusion_layer=6,
        # encoder_width=768)), so the checkpoint carries a full-size temporal_encoder
        from videotgb_amd import synth
        cfg.tgb = synth.TgbCfg()
        sd = full_state_dict(cfg, models.build_language_model(models.load_hf_config(base, "instructblip")))
        ckpt = str(tmp_path / "last.ckpt")
        torch.save({"state_dict": sd}, ckpt)
        bu.AutoProcessor = types.SimpleNamespace(from_pretrained=lambda *a, **k: "processor")
        bu.AutoTokenizer = types.SimpleNamespace(from_pretrained=lambda *a, **k: "sampler_processor")
        model, proc, sproc = bu.load_pretrained_model(ckpt, base, "bert-base-uncased", "cpu", False)
        assert type(model) is models.LSTP and (proc, sproc) == ("processor", "sampler_processor")
        got = model.state_dict()
        for k in ("model.qformer.encoder.layer.0.attention.attention.query.weight", "temporal_encoder.mrc_head.weight",
                  "of_extractor.update_block.gru.convz1.weight", "model.language_model.lm_head.weight"):
            assert torch.equal(got[k], sd[k].to(got[k].dtype)), k       # (the LLM is built in bf16 with the default compute dtype)
        assert hasattr(model, "generate") and model.model.config.use_decoder_only_language_model
    finally:
        dropin.uninstall()
        sys.path.remove(REF)
        for name in stubbed:
            sys.modules.pop(name, None)
        for name in [n for n in sys.modules if n == "eval" or n.startswith("eval.") or n == "src" or n.startswith("src.")]:
            sys.modules.pop(name, None)
