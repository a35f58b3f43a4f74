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


def _inputs(g, n, Cin, H, W):
    x = torch.randn(n, Cin, H, W, generator=g) * (torch.rand(n, 1, H, W, generator=g) * 10 + 0.1)
    return torch.relu(x) + 0.05 * torch.randn(n, Cin, H, W, generator=g)


@pytest.mark.parametrize("N,C1,obf,H,W,n", [(64, 64, False, 40, 40, 2), (64, 64, True, 17, 23, 3), (128, 128, False, 28, 28, 3), (128, 128, True, 16, 16, 5),
                                          (96, 128, False, 28, 28, 2)])
def test_residual_tail_in_the_epilogue_vs_fp64(dev, N, C1, obf, H, W, n):
    """The ResidualBlock tail relu(x + relu(conv(y) + b)) (extractor.py:56-60) in the pair-store epilogue of the 64- and 128-wide tiles (cnet's blocks at
    f16c8; vtgb_pair_conv_ex resid): against fp64 from the SAME rounded operands; N = 96 in 128-channel rows: the padded channels come out as zeros."""
    from videotgb_amd import ops
    g = torch.Generator().manual_seed(N * 3 + C1 + H)
    y = _inputs(g, n, C1, H, W)
    xs = _inputs(g, n, N, H, W)
    w = torch.randn(N, C1, 3, 3, generator=g) * 0.05
    b = torch.randn(N, generator=g)
    ld = C1
    skip_rows = ops.pair_pack(xs.permute(0, 2, 3, 1).reshape(-1, N).contiguous().to(dev), ops.F16C8, ld)
    skip = ops.pair_unpack(skip_rows, N).cpu().view(n, H, W, N).permute(0, 3, 1, 2).double()                 # what the kernel adds: the pair's value
    ref = (skip + F.conv2d(y.double(), w.double(), b.double(), padding=1).relu()).relu()
    bound = F.conv2d(y.abs().double(), w.abs().double(), None, padding=1) + b.abs().double().view(1, -1, 1, 1) + skip.abs()
    a = ops.pair_pack(y.permute(0, 2, 3, 1).reshape(-1, C1).contiguous().to(dev))
    wk = w.permute(0, 2, 3, 1).contiguous().to(dev)
    sw, _ = ops.h8_weight_scale(wk)
    ofmt = ops.BF16X3 if obf else ops.F16C8
    out = ops.pair_conv(a, wk, sw, H, W, bias=b.to(dev), relu=True, out_fmt=ofmt, ld_out=ld, resid=skip_rows)
    got = ops.pair_unpack(out, N, ofmt).cpu().view(n, H, W, N).permute(0, 3, 1, 2).double()
    err = ((got - ref).abs() / bound).max().item()
    print(f"[residual tail N={N} C1={C1} bf16-pair-out={obf}] max err / bound = {err:.3e}")
    assert err <= 2.0 ** -14
    if N < ld:
        assert not out.cpu()[:, N:ld].any() and not out.cpu()[:, ld + N:].any()


@pytest.mark.parametrize("H,W,n", [(28, 28, 3), (9, 13, 7)])
def test_flow_head_tail_vs_fp64(dev, H, W, n):
    """FlowHead (update.py:10-18): conv1 (3x3, 128 -> 256, ReLU) with conv2's 18 per-tap products formed in its epilogue on the fp32 matrix instruction
    (gemm_h8.hip EPI_FTAIL; vtgb_pair_conv_ex tail_w) against fp64: hidden . W2 is EXACT fp32 here, so the error is the convolution's own."""
    from videotgb_amd import ops
    g = torch.Generator().manual_seed(H * 31 + n)
    h = torch.tanh(torch.randn(n, 128, H, W, generator=g))
    w1 = torch.randn(256, 128, 3, 3, generator=g) * 0.04
    b1 = torch.randn(256, generator=g) * 0.1
    w2 = torch.zeros(32, 256)
    w2[:18] = torch.randn(18, 256, generator=g) * 0.05
    hidden = F.conv2d(h.double(), w1.double(), b1.double(), padding=1).relu()                       # [n, 256, H, W]
    ref = torch.einsum("nchw,oc->nhwo", hidden, w2.double()).reshape(-1, 32)
    hb = F.conv2d(h.abs().double(), w1.abs().double(), None, padding=1) + b1.abs().double().view(1, -1, 1, 1)
    bound = torch.einsum("nchw,oc->nhwo", hb, w2.abs().double()).reshape(-1, 32)
    a = ops.pair_pack(h.permute(0, 2, 3, 1).reshape(-1, 128).contiguous().to(dev))
    wk = w1.permute(0, 2, 3, 1).contiguous().to(dev)
    sw, _ = ops.h8_weight_scale(wk)
    got = ops.pair_conv(a, wk, sw, H, W, bias=b1.to(dev), relu=True, tail_w=w2).cpu().double()
    err = ((got[:, :18] - ref[:, :18]).abs() / bound[:, :18]).max().item()
    print(f"[flow-head tail {n} x {H} x {W}] max err / bound = {err:.3e}")
    assert err <= 2.0 ** -14 and not got[:, 18:].any()
    again = ops.pair_conv(a, wk, sw, H, W, bias=b1.to(dev), relu=True, tail_w=w2).cpu().double()
    assert torch.equal(got, again)                                                                  # the four column waves' sums meet in a fixed order


@pytest.mark.parametrize("N,C1,H,W,n", [(64, 64, 40, 40, 2), (96, 128, 28, 28, 3), (128, 128, 16, 16, 5)])
def test_fp32_output_tiles_vs_fp64(dev, N, C1, H, W, n):
    """fp32 rows straight from the accumulators (fnet's convolutions in front of an InstanceNorm: 64- and 128-wide tiles; vtgb_pair_conv_ex out_f32)."""
    from videotgb_amd import ops
    g = torch.Generator().manual_seed(N + C1 + W)
    y = _inputs(g, n, C1, H, W)
    w = torch.randn(N, C1, 3, 3, generator=g) * 0.05
    b = torch.randn(N, generator=g)
    ref = F.conv2d(y.double(), w.double(), b.double(), padding=1)
    bound = F.conv2d(y.abs().double(), w.abs().double(), None, padding=1) + b.abs().double().view(1, -1, 1, 1)
    a = ops.pair_pack(y.permute(0, 2, 3, 1).reshape(-1, C1).contiguous().to(dev))
    wk = w.permute(0, 2, 3, 1).contiguous().to(dev)
    sw, _ = ops.h8_weight_scale(wk)
    out = ops.pair_conv(a, wk, sw, H, W, bias=b.to(dev), ld_out=C1, out_f32=True).cpu()
    got = out[:, :N].view(n, H, W, N).permute(0, 3, 1, 2).double()
    err = ((got - ref).abs() / bound).max().item()
    print(f"[fp32 rows N={N} C1={C1}] max err / bound = {err:.3e}")
    assert err <= 2.0 ** -14
