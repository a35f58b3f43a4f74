"""-m gpu: the VTGB_F16C8 operand format piece by piece (csrc/pair_h8.h, gemm_h8.hip; include/vtgb.h vtgb_pair_pack / vtgb_pair_conv): the device's
pair rows against the CPU statement of the format byte for byte, and single convolutions of the update block's shapes against fp64 convolutions.
Bounds: a pair carries ~15-16 significant bits of each operand, so a convolution is within 2^-14 of sum |x| |w| (observed ~2^-16); the plain fp16
product is ~2^-11."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from videotgb_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def test_pair_rows_match_the_format_byte_for_byte(dev):
    from test_oracle import _h8_pack_rows
    from videotgb_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(300, 128, generator=g) * torch.logspace(-6, 5, 300).unsqueeze(1)      # 1e-6 .. 1e5: subnormal residuals up to saturation
    x[0, :8] = torch.tensor([0.0, -0.0, 57344.0, 60000.0, -1e9, 6.1e-5, 3e-8, 65504.0])
    got = ops.pair_pack(x.to(dev), ops.F16C8).cpu().view(torch.uint8)
    want = _h8_pack_rows(x)
    assert torch.equal(got, want)
    back = ops.pair_unpack(ops.pair_pack(x.to(dev)), 128).cpu()
    xc = x.clamp(-57344.0, 57344.0)
    assert ((back - xc).abs() <= 2.0 ** -14 * xc.abs() + 2.0 ** -25).all()
    # padded rows: channels [C, ld) are zeros in both halves
    p = ops.pair_pack(x[:, :100].contiguous().to(dev), ops.F16C8, 128).cpu()
    assert torch.equal(p[:, 100:128], torch.zeros(300, 28, dtype=torch.int16)) and torch.equal(p[:, 128 + 100:], torch.zeros(300, 28, dtype=torch.int16))


CASES = [  # (N, KH, KW, C1, two sources, out as bf16 pair, H, W, images)
    (256, 1, 1, 384, False, False, 28, 28, 3),      # convc1
    (192, 3, 3, 256, False, False, 28, 28, 2),      # convc2: the 256 x 192 tile
    (126, 3, 3, 256, False, False, 16, 16, 3),      # the motion convolution: N % 4 == 2
    (256, 3, 3, 128, False, True, 28, 28, 2),       # flow_head.conv1: bf16 pair out
    (128, 1, 5, 128, True, False, 9, 13, 5),        # a two-source horizontal convolution on tiles that straddle images
    (256, 5, 1, 128, True, False, 28, 28, 1),       # vertical, two sources, 256 wide
]


@pytest.mark.parametrize("N,KH,KW,C1,two,obf,H,W,n", CASES)
def test_pair_conv_vs_fp64(dev, N, KH, KW, C1, two, obf, H, W, n):
    from videotgb_amd import ops
    g = torch.Generator().manual_seed(N + KH * 7 + C1)
    Cin = C1 * (2 if two else 1)
    x = torch.randn(n, Cin, H, W, generator=g) * (torch.rand(n, 1, H, W, generator=g) * 30 + 0.1)
    x = torch.relu(x) + 0.05 * torch.randn(n, Cin, H, W, generator=g)            # post-ReLU-like: mostly positive, wide range of magnitudes
    w = torch.randn(N, Cin, KH, KW, generator=g) * 0.05
    b = torch.randn(N, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=(KH // 2, KW // 2)).relu()
    bound = F.conv2d(x.abs().double(), w.abs().double(), None, padding=(KH // 2, KW // 2)) + b.abs().double().view(1, -1, 1, 1)
    rows = x.permute(0, 2, 3, 1).reshape(n * H * W, Cin)
    a = ops.pair_pack(rows[:, :C1].contiguous().to(dev))
    a2 = ops.pair_pack(rows[:, C1:].contiguous().to(dev)) if two else None
    wk = w.permute(0, 2, 3, 1).contiguous().to(dev)
    sw, _ = ops.h8_weight_scale(wk)
    ofmt = ops.BF16X3 if obf else ops.F16C8
    out = ops.pair_conv(a, wk, sw, H, W, a2=a2, bias=b.to(dev), relu=True, out_fmt=ofmt)
    got = ops.pair_unpack(out, N, ofmt).cpu().view(n, H, W, N).permute(0, 3, 1, 2).double()
    err = ((got - ref).abs() / bound).max().item()
    x16 = x.to(torch.float16).double()
    err16 = ((F.conv2d(x16, w.to(torch.float16).double(), b.double(), padding=(KH // 2, KW // 2)).relu() - ref).abs() / bound).max().item()
    print(f"[pair conv N={N} {KH}x{KW} C1={C1} two={two}] max err / sum|x||w| = {err:.3e} (2^-14 = {2.0 ** -14:.3e}); fp16-only: {err16:.3e}")
    assert err <= 2.0 ** -14 and err16 > 4 * err


@pytest.mark.parametrize("N,KH,KW,C1,two,obf", [(256, 3, 3, 128, False, True), (256, 1, 5, 128, True, False), (192, 3, 3, 256, False, False)])
def test_pair_conv_wide_and_narrow_tiles_agree_bit_for_bit(dev, N, KH, KW, C1, two, obf):
    """launch_conv_h8 gives a 256-channel convolution the 256-wide tile on large batches and two 128-wide n-tiles on few m-tiles (a single clip);
    the contraction order of an output element is the same in both, so 400 images in one call (1 225 m-tiles: wide) and the first two of them alone
    (7 m-tiles: narrow) must agree bit for bit -- which also carries the fp64 check of the small cases above over to the wide instantiations."""
    from videotgb_amd import ops
    g = torch.Generator(device=dev).manual_seed(N + KW)
    n, H, W = 400, 28, 28
    x = torch.randn(n * H * W, C1, generator=g, device=dev).abs() * 3
    a = ops.pair_pack(x)
    a2 = ops.pair_pack(torch.randn(n * H * W, C1, generator=g, device=dev)) if two else None
    wk = torch.randn(N, KH, KW, C1 * (2 if two else 1), generator=g, device=dev) * 0.05
    sw, _ = ops.h8_weight_scale(wk)
    ofmt = ops.BF16X3 if obf else ops.F16C8
    full = ops.pair_conv(a, wk, sw, H, W, a2=a2, relu=True, out_fmt=ofmt)
    m2 = 2 * H * W
    part = ops.pair_conv(a[:m2].contiguous(), wk, sw, H, W, a2=None if a2 is None else a2[:m2].contiguous(), relu=True, out_fmt=ofmt)
    assert torch.equal(full[:m2], part) and full.abs().max() > 0
