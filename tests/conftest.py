import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: (torch.from_numpy(z[k]) if z[k].ndim else z[k].item()) for k in z.files}


def deq(g, key, scale_key="q8_scale"):
    return g[key].float() * float(g[scale_key])


@pytest.fixture(scope="session")
def tiny_sd():
    from videotgb_amd.synth import path_state_dict, tiny_cfg
    out = {}
    for arch in ("instructblip", "blip2"):
        cfg = tiny_cfg(arch)
        cfg.vit.image = 56
        out[arch] = (cfg, path_state_dict(cfg, seed=0))
    return out
