"""CPU-side checks of the C ABI: the library builds, loads and exports every symbol that
include/vtgb.h declares; host-side argument validation returns the documented codes."""
import ctypes as C
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from videotgb_amd import build
    build.build()
    from videotgb_amd import _lib
    return _lib


def test_exports_match_header(lib):
    hdr = open(os.path.join(REPO, "include", "vtgb.h")).read()
    declared = set(re.findall(r"\b(vtgb_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"vtgb_stream_t"}
    assert declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
    L = lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.vtgb_version() == 601


def test_struct_sizes_follow_header(lib):
    # field-by-field mirror of the header; the C side is compiled from the same header, so a
    # size agreement on the biggest structs is a cheap layout check
    assert C.sizeof(lib.VitArgs) == 8 * 4 + 4 + 4 + 6 * 8
    assert C.sizeof(lib.GemmArgs) == 5 * 4 + 4 + 8 * 8
    assert C.sizeof(lib.SpanSelectArgs) == 3 * 8 + 4 * 4


def test_argument_validation_without_gpu(lib):
    L = lib.lib()
    a = lib.SpanSelectArgs(None, None, None, 1, 4, 2, 0.5)
    assert L.vtgb_span_select(C.byref(a), None) == -1
    assert b"NULL" in L.vtgb_last_error()
    with pytest.raises(ValueError):
        lib.check(-1)
    v = lib.VitArgs(lib.BF16, 8, 224, 14, 1408, 16, 6144, 39, 1e-6, None, None, None, None, None, 0)
    need = L.vtgb_vit_workspace_bytes(C.byref(v))
    assert 50e6 < need < 200e6
    assert L.vtgb_vit_forward(C.byref(v), None) == -2          # workspace missing
    t = lib.TgbArgs(lib.BF16, 1, 96, 14, 768, 12, 3072, 12, 6, 7, 224, 16, 1e-12, None, None, None, None, None, None, None, None, 0)
    assert L.vtgb_tgb_workspace_bytes(C.byref(t)) == 0          # invalid mode
    assert b"INVALID MODE" in L.vtgb_last_error()
    assert L.vtgb_vit_patch_kpad(lib.BF16, 14) == 640 and L.vtgb_vit_patch_kpad(lib.F32, 14) == 588


def test_argument_validation_of_the_wider_rows(lib):
    """RAFT, preprocessing, training-loss and pyramid entry points reject bad arguments on the host (no GPU needed)."""
    L = lib.lib()
    EINVAL, EWS = -1, -2
    assert C.sizeof(lib.RaftUpdateArgs) == 5 * 4 + 4 + 2 * 8 + 4 * 8 + 4 * 8 + 8 + 8 + 8                # dtype first; corr_f16 (+pad), cnet_nhwc, flow_init
    r = lib.RaftUpdateArgs(lib.BF16, 0, 28, 28, 20, None, None, (C.c_void_p * 4)(), None, None, None, 0, 1, None, None)
    assert L.vtgb_raft_update_workspace_bytes(C.byref(r)) == 0 and b"bad dims" in L.vtgb_last_error()
    r.n_pairs = 95
    need = L.vtgb_raft_update_workspace_bytes(C.byref(r))
    assert 0.4e9 < need < 1.2e9                                                                        # ~7 KB per coarse pixel
    assert L.vtgb_raft_update(C.byref(r), None) == EWS
    r.dtype = 7
    assert L.vtgb_raft_update_workspace_bytes(C.byref(r)) == 0 and b"bad dtype" in L.vtgb_last_error()
    r.dtype = lib.F32                                                                                  # exactness mode: fp32 activations
    assert need < L.vtgb_raft_update_workspace_bytes(C.byref(r)) < 2 * need
    e = lib.RaftEncoderArgs(lib.BF16, 4, 60, 224, 0, None, None, None, None, 0)
    assert L.vtgb_raft_encoder_workspace_bytes(C.byref(e)) == 0 and b"bad dims" in L.vtgb_last_error()
    e = lib.RaftEncoderArgs(lib.F32, 4, 224, 224, 0, None, None, None, None, 0)
    assert L.vtgb_raft_encoder_workspace_bytes(C.byref(e)) > 0 and L.vtgb_raft_encoder(C.byref(e), None) == EWS
    pp = lib.PreprocessArgs(None, None, None, 4, 240, 320, 4, 224, (C.c_float * 3)(0, 0, 0), (C.c_float * 3)(1, 1, 1))
    assert L.vtgb_preprocess_frames(C.byref(pp), None) == EINVAL and b"NULL" in L.vtgb_last_error()
    ct = lib.ConcatTextIoArgs(None, None, None, None, None, None, None, None, 0, 2, 8, 4, 32)
    assert L.vtgb_concat_text_io(C.byref(ct), None) == EINVAL
    ce = lib.ShiftedCeArgs(lib.BF16, 2, 1, 32000, None, None, None, None, None, None, None)
    assert L.vtgb_shifted_ce_forward(C.byref(ce), None) == EINVAL and L.vtgb_shifted_ce_backward(C.byref(ce), None) == EINVAL
    cp = lib.RaftCorrArgs(lib.BF16, 95, 28, 28, 256, 95, 96, 0, 1, 96, 1 / 16.0, None, (C.c_void_p * 4)(), None, 0)
    assert L.vtgb_raft_corr_workspace_bytes(C.byref(cp)) == 96 * 784 * 256 * 2                          # the fp16 copy of the feature maps
    assert L.vtgb_raft_corr(C.byref(cp), None) == EINVAL and b"NULL" in L.vtgb_last_error()
    cp.n_images = 95                                                                                   # the last pair would read image 95
    assert L.vtgb_raft_corr_workspace_bytes(C.byref(cp)) == 0 and b"pair -> image map" in L.vtgb_last_error()
    cp.n_images, cp.dim = 96, 128
    assert L.vtgb_raft_corr(C.byref(cp), None) == EINVAL and b"bad dims" in L.vtgb_last_error()
    # decode-step GEMM: M <= 128, K a multiple of 64; one fp32 fragment per (128-column tile, split)
    sk = lib.GemmSkinnyArgs(124, 4096, 4096, 0, None, 4096, None, 4096, None, 4096, lib.BF16, 0, None, 0)
    need = L.vtgb_gemm_skinny_workspace_bytes(C.byref(sk))
    assert need % (32 * 124 * 128 * 4) == 0 and 2 <= need // (32 * 124 * 128 * 4) <= 8                  # 32 tiles x 2..8 splits
    assert L.vtgb_gemm_skinny(C.byref(sk), None) == EINVAL and b"NULL" in L.vtgb_last_error()
    sk.n_splits = 3
    assert L.vtgb_gemm_skinny_workspace_bytes(C.byref(sk)) == 3 * 32 * 124 * 128 * 4
    sk.n_splits = 1                                                                                    # no split: stored straight to `out`
    assert L.vtgb_gemm_skinny_workspace_bytes(C.byref(sk)) == 0
    sk.M = 129
    assert L.vtgb_gemm_skinny_workspace_bytes(C.byref(sk)) == 0 and b"M=129" in L.vtgb_last_error()
    sk.M, sk.K = 1, 100
    assert L.vtgb_gemm_skinny_workspace_bytes(C.byref(sk)) == 0 and b"K=100" in L.vtgb_last_error()
    assert L.vtgb_pack_skinny_weight_bytes(32000, 4096) == 250 * 64 * 16384 and L.vtgb_pack_skinny_weight_bytes(100, 100) == 0
    assert L.vtgb_pack_skinny_weight(None, 4096, 4096, 4096, None, None) == EINVAL


def test_product_path_has_no_cpu_fallback(lib):
    import torch
    from videotgb_amd import ops
    with pytest.raises(lib.VtgbError, match="no CPU"):
        ops.span_select(torch.zeros(1, 4, 2), torch.zeros(2, 2, 4))


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "videotgb_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+\S*oracle", src, re.M), f
