"""TEST INFRASTRUCTURE (uses the oracle as the checker; not collected by pytest, not product).  CPU emulation of candidate split-operand schemes for RAFT's convolutions (no GPU): which operand formats keep the flows of the
input-sensitive weight set at 224 x 224 / 20 iterations within the bf16x3 mode's bound of the fp32 oracle?

    python tests/emul_f16c8.py [scheme ...]      schemes: bf16, bf16x3, f16, f16x2, f16c8, f16c8r (what the device does: e5m2 activations, no
                                                 data-dependent scale), or a per-part map "m:update=f16c8r,fnet=bf16x3,cnet=bf16x3,layer1=f16c8r"
                                                 (the LAST key contained in a convolution's name wins; the product's configuration is that one)

bf16x3 : x ~ hi + lo (bf16), w ~ Wh + Wl (bf16): hi.Wh + lo.Wh + hi.Wl                  (the round-5 mode)
f16c8  : x ~ xh (fp16) + xl, main product xh.Wh in fp16; the two corrections xl.Wh and xh.Wl with BOTH operands in OCP e4m3 (power-of-two
         scales): they are 2^-11 of the product, so 4 significant bits on each side leave ~2^-16.
Products are summed in fp64 here (the question is the operand formats, not the accumulation order)."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import vtgb_oracle as O          # noqa: E402
from videotgb_amd import synth               # noqa: E402

E4M3_MAX = 448.0
stats = {}


def e4m3(x):
    return x.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).to(torch.float64)


def e5m2(x):
    return x.clamp(-57344.0, 57344.0).to(torch.float8_e5m2).to(torch.float64)


def f16(x):
    return x.clamp(-65504.0, 65504.0).to(torch.float16).to(torch.float64)


def bf16(x):
    return x.to(torch.bfloat16).to(torch.float64)


def pow2_scale(t, target):
    m = float(t.abs().max())
    if m == 0.0:
        return 1.0
    import math
    return 2.0 ** math.floor(math.log2(target / m))


def make_conv(scheme, only_update):
    scheme0 = scheme

    def conv(sd, name, x, stride=1, padding=0):
        w, b = sd[name + ".weight"], sd[name + ".bias"]
        scheme = scheme0
        if ":" in scheme0:      # per-part map "update=f16c8r,fnet=bf16x3,cnet=bf16[,layer1=...]": the LAST matching key wins
            scheme = "f32"
            for kv in scheme0.split(":")[1].split(","):
                k, v = kv.split("=")
                if k in name:
                    scheme = v
        if scheme == "f32" or (only_update and "update_block" not in name) or name.endswith("convf1"):
            return F.conv2d(x, w, b, stride=stride, padding=padding)
        xd, wd = x.double(), w.double()
        cv = lambda a, ww: F.conv2d(a, ww, None, stride=stride, padding=padding)
        st = stats.setdefault(name, [0.0, 0.0])
        st[0] = max(st[0], float(x.abs().max())); st[1] = max(st[1], float(w.abs().max()))
        if scheme == "bf16":
            y = cv(bf16(x), bf16(w))
        elif scheme == "bf16x3":
            xh = bf16(x); xl = bf16(xd - xh); wh = bf16(w); wl = bf16(wd - wh)
            y = cv(xh, wh) + cv(xl, wh) + cv(xh, wl)
        elif scheme == "f16":
            y = cv(f16(x), f16(w))
        elif scheme == "f16x2":          # x to 22 bits, w to 11
            xh = f16(x); xl = f16(xd - xh); wh = f16(w)
            y = cv(xh, wh) + cv(xl, wh)
        elif scheme in ("f16c8r", "mixed"):
            # the robust form: activations' correction operands in e5m2 WITHOUT a data-dependent scale (e5m2 = fp16's range), weights in e4m3
            # with a per-layer power-of-two scale fixed at pack time.  "mixed": this in the update block, bf16x3 in the encoders.
            if scheme == "mixed" and "update_block" not in name:
                xh = bf16(x); xl = bf16(xd - xh); wh = bf16(w); wl = bf16(wd - wh)
                y = cv(xh, wh) + cv(xl, wh) + cv(xh, wl)
            else:
                xh = f16(x); xl = xd - xh; wh = f16(w); wl = wd - wh
                sw = pow2_scale(w, E4M3_MAX)
                xl8 = e5m2(xl * 4096.0); xh8 = e5m2(xd)
                wh8 = e4m3(wd * sw); wl8 = e4m3(wl * sw * 4096.0)
                y = cv(xh, wh) + (cv(xl8, wh8) + cv(xh8, wl8)) / (sw * 4096.0)
        elif scheme in ("f16c8", "f16c8fix"):
            xh = f16(x); xl = xd - xh; wh = f16(w); wl = wd - wh
            # weight scale: per layer, from max |w| (pack time); activation scale: per call (f16c8: what a perfect scale would give) or fixed 2^4
            sw = pow2_scale(w, E4M3_MAX)
            sx = pow2_scale(x, E4M3_MAX) if scheme == "f16c8" else FIX.get(name.split(".")[-1], 16.0)
            xl8 = e4m3(xl * sx * 2048.0); xh8 = e4m3(xd * sx)
            wh8 = e4m3(wd * sw); wl8 = e4m3(wl * sw * 2048.0)
            y = cv(xh, wh) + (cv(xl8, wh8) + cv(xh8, wl8)) / (sx * sw * 2048.0)
        else:
            raise ValueError(scheme)
        return (y + b.double().view(1, -1, 1, 1)).float()
    return conv


FIX = {}


def rel_rms(a, b):
    return float(((a - b).double().pow(2).mean() / b.double().pow(2).mean()).sqrt())


def main():
    schemes = sys.argv[1:] or ["bf16", "bf16x3", "f16", "f16x2", "f16c8", "f16c8fix"]
    torch.set_num_threads(8)
    sd = synth.raft_sensitive_state_dict(0)
    frames = synth.clip_normalise(synth.moving_texture_u8(3, 224, 7))
    img = (frames * 0.5 + 0.5) * 255.0 if frames.abs().max() < 4 else frames
    orig = O._conv
    ref = O.raft_forward(sd, "of_extractor.", frames[:-1], frames[1:], 20)
    print(f"reference: max|flow| = {float(ref.abs().max()):.3f}")
    for only_update in ((False,) if any(":" in s for s in schemes) else (True, False)):
        for s in schemes:
            O._conv = make_conv(s, only_update)
            got = O.raft_forward(sd, "of_extractor.", frames[:-1], frames[1:], 20)
            O._conv = orig
            print(f"{'update block only' if only_update else 'all convolutions  '} {s:60s} flow rel_rms vs fp32 = {rel_rms(got, ref):.3e}   max|diff| = {float((got - ref).abs().max()):.3e}", flush=True)
    for k, v in stats.items():
        print(f"  {k:55s} max|x| {v[0]:10.3f}  max|w| {v[1]:8.4f}")


if __name__ == "__main__":
    main()
