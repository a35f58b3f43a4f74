"""Drop-in registration: make the reference's own drivers resolve to the HIP-backed twins WITHOUT editing them.

    import videotgb_amd.dropin; videotgb_amd.dropin.install()     # one line, before the reference's entry point runs
    # ... then, unchanged:
    #   python -m eval.inference --model_path ckpt --model_base <instructblip dir> ...     (eval/inference.py)
    #   python src/train.py experiment=...                                                  (src/train.py:53 hydra instantiate)

``install()`` pre-registers in ``sys.modules``
  * ``eval.utils.model`` with ``LSTP`` / ``LSTP_blip2`` = videotgb_amd.models' classes (same constructor
    ``(base_model_path, device, lora)``, same ``generate`` signature, same state_dict keys), so that the reference's own
    ``eval/utils/builder_utils.load_pretrained_model`` -- which does ``from .model import LSTP, LSTP_blip2`` -- builds and
    loads the HIP-backed model;
  * the modules named by the Hydra ``_target_``s of configs/model/*.yaml (``src.models.LSTP_module`` ...) with the twins
    of videotgb_amd.modules.
Python's import system returns a ``sys.modules`` entry before looking at the file system, so the reference's files of the
same dotted name are simply never executed; everything else of the reference (datamodules, conversation templates,
launch scripts, Lightning, Hydra) keeps running as it is.  ``uninstall()`` removes the entries again.
"""
from __future__ import annotations

import importlib.machinery
import sys
import types
from typing import Dict, List

_installed: List[str] = []


def _module(name: str, attrs: Dict[str, object]) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__videotgb_amd_dropin__ = True
    return m


def install(eval_model: bool = True, lightning_modules: bool = True) -> List[str]:
    """Register the twins under the reference's module paths; returns the names registered."""
    from . import builder_utils, models, modules
    names: List[str] = []
    if eval_model:
        # eval/utils/builder_utils.py:14 imports LSTP, LSTP_blip2 from .model -- give it ours; the parent packages are left
        # to the normal import system (they exist on disk in the reference checkout)
        names.append("eval.utils.model")
        sys.modules["eval.utils.model"] = _module("eval.utils.model", {
            "LSTP": models.LSTP, "LSTP_blip2": models.LSTP_blip2, "InputPadder": models.InputPadder, "RAFT": models.Raft})
    if lightning_modules:
        by_mod: Dict[str, Dict[str, object]] = {}
        for target, cls in modules.TARGETS.items():
            mod, attr = target.rsplit(".", 1)
            by_mod.setdefault(mod, {})[attr] = cls
        for mod, attrs in by_mod.items():
            names.append(mod)
            sys.modules[mod] = _module(mod, attrs)
    _installed.extend(names)
    return names


def uninstall() -> None:
    for n in _installed:
        m = sys.modules.get(n)
        if m is not None and getattr(m, "__videotgb_amd_dropin__", False):
            del sys.modules[n]
    _installed.clear()
