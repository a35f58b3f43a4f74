#!/usr/bin/env python3
"""Every BASELINE.json config once, at FULL size, through the reference-shaped entry points (random-init weights, synthetic
data).  One line per config: shape facts, clips/s (or sequences/s) of a short timed loop.  C3 is bench.py's default and C5 is
tools/train_bench.py; they are run from here as sub-steps so that all five lines come from one command:

    python tools/configs_check.py [c1 c2 c3 c4 c5]

C1  BLIP2-Flan-T5-xl, no TGB sampler (range(32) -> midpoint subsample to 4 frames), greedy   modules.LSTPBlip2Module.eval_forward
C2  BLIP2-Flan-T5-xl + TGB (fusion, precomputed flow of length 32, map B), 32 -> 8           modules.LSTPSFBlip2Module.eval_forward
C3  InstructBLIP-Vicuna-7B + TGB, RAFT inline, T = 96 -> 8                                   models.LSTP.generate (bench.py)
C4  same, T = 256 -> 8                                                                       models.LSTP.generate
C5  Vicuna-7B LoRA + Q-Former training micro-step                                            train.LoraTrainStep (tools/train_bench.py)
(C1's "runs without a GPU" is not offered: the product has no CPU path.)"""
import functools
import os
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


class BE(dict):
    __getattr__ = dict.__getitem__


def write_blip2_t5xl(dirname):
    from transformers import Blip2Config, Blip2QFormerConfig, Blip2VisionConfig, T5Config
    tc = T5Config(vocab_size=32128, d_model=2048, d_kv=64, d_ff=5120, num_layers=24, num_decoder_layers=24, num_heads=32,
                  feed_forward_proj="gated-gelu", tie_word_embeddings=False, decoder_start_token_id=0, pad_token_id=0, eos_token_id=1,
                  architectures=["T5ForConditionalGeneration"])
    c = Blip2Config(vision_config=Blip2VisionConfig().to_dict(), qformer_config=Blip2QFormerConfig().to_dict(), text_config=tc.to_dict(),
                    num_query_tokens=32)
    os.makedirs(dirname, exist_ok=True)
    c.save_pretrained(dirname)
    return dirname


def timed(fn, n=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n


def blip2_module(cls_name, tmp, dev):
    from videotgb_amd import modules, synth
    base = write_blip2_t5xl(os.path.join(tmp, "blip2-flan-t5-xl"))
    cfg = synth.full_cfg("blip2")
    sd = synth.path_state_dict(cfg, 0, with_raft=True)
    raft = os.path.join(tmp, "raft-things.pth")
    torch.save({"module." + k[len("of_extractor."):]: v for k, v in sd.items() if k.startswith("of_extractor.")}, raft)
    proc = BE(tokenizer=BE(pad_token_id=0), batch_decode=lambda ids, skip_special_tokens=True: [""] * len(ids))
    m = getattr(modules, cls_name)(model_name_or_path=base, sampler_name_or_path=os.path.join(tmp, "none"), of_extractor_name_or_path=raft,
                                   temperature=1.0, optimizer=functools.partial(torch.optim.AdamW, lr=1e-4), scheduler=None, scheduler_params={},
                                   generate_configs=dict(do_sample=False, max_new_tokens=16, min_new_tokens=16), compute_dtype="bf16", processor=proc)
    m.load_state_dict(sd, strict=False)
    return m.to(dev).eval()


def graph_leg(m, batch, out_hf, reps):
    """The same eval_forward with the module's opt-in graph decoder (modules.fast_decode): same shape, same first generated token
    (bf16, random weights: later greedy tokens can part from HF's at near ties -- the fp32 id-for-id check is tests/test_decode.py)."""
    m.fast_decode = True
    try:
        out = m.eval_forward(batch)
        assert tuple(out.shape) == tuple(out_hf.shape), (tuple(out.shape), tuple(out_hf.shape))
        assert out[:, :2].tolist() == out_hf[:, :2].tolist()
        assert getattr(m, "_graph_decoder", None) is not None
        return timed(lambda: m.eval_forward(batch), n=reps)
    finally:
        m.fast_decode = False


def c1(dev, tmp, B=32, reps=3, module=None):
    m = module if module is not None else blip2_module("LSTPBlip2Module", tmp, dev)
    nframe = 4
    g = torch.Generator(device=dev).manual_seed(1)
    batch = dict(frames=torch.randn(B * 32, 3, 224, 224, generator=g, device=dev), nframe=nframe, of_lengths=[32] * B,
                 answer=torch.zeros(B, 1, dtype=torch.long, device=dev), text_answer=[""] * B,
                 question=torch.randint(3, 32000, (B, 20), generator=g, device=dev), question_attention_mask=torch.ones(B, 20, dtype=torch.long, device=dev))
    out, st = m.eval_forward(batch, return_stages=True)
    assert st["frame_idx"][0].tolist() == [3, 11, 19, 27] and st["of_logits"] is None, st["frame_idx"][0].tolist()
    assert tuple(out.shape) == (B, 17) or tuple(out.shape) == (B, 16), tuple(out.shape)      # 16 greedy tokens (+ the decoder start token)
    assert tuple(st["language_model_inputs"].shape) == (B, nframe * 32, 2048)
    dt_hf = timed(lambda: m.eval_forward(batch), n=reps)
    dt = graph_leg(m, batch, out, reps)
    line = (f"C1 BLIP2-Flan-T5-xl, no sampler, 32 -> {nframe} frames (idx {st['frame_idx'][0].tolist()}), concat prefix {tuple(st['language_model_inputs'].shape)}, "
            f"T5 greedy 16 tokens: {B / dt:.1f} clips/s at {B} clips per call (T5GreedyDecoder, hipGraph; HF generate, eager: {B / dt_hf:.1f})")
    print(line)
    return dict(clips_per_s=B / dt, frame_idx=st["frame_idx"][0].tolist(), line=line)


def c2(dev, tmp, B=32, reps=3, module=None):
    m = module if module is not None else blip2_module("LSTPSFBlip2Module", tmp, dev)
    nframe, L = 8, 32
    g = torch.Generator(device=dev).manual_seed(2)
    batch = dict(frames=torch.randn(B * 32, 3, 224, 224, generator=g, device=dev), nframe=nframe, of_lengths=[L] * B,
                 of=torch.rand(B, L, 2, 224, 224, generator=g, device=dev) * 2 - 1, of_mask=torch.ones(B, L + 2, dtype=torch.long, device=dev),
                 sampler_question=torch.randint(1000, 30000, (B, 14), generator=g, device=dev),
                 sampler_question_attention_mask=torch.ones(B, 14, dtype=torch.long, device=dev),
                 answer=torch.zeros(B, 1, dtype=torch.long, device=dev), text_answer=[""] * B,
                 question=torch.randint(3, 32000, (B, 20), generator=g, device=dev), question_attention_mask=torch.ones(B, 20, dtype=torch.long, device=dev))
    out, st = m.eval_forward(batch, return_stages=True)
    assert st["of_logits"].shape == (B, L, 2)
    assert tuple(st["language_model_inputs"].shape) == (B, nframe * 32, 2048) and tuple(st["frame_idx"].shape) == (B, nframe)
    assert bool((st["frame_idx"][:, 1:] >= st["frame_idx"][:, :-1]).all()) and int(st["frame_idx"].max()) < 32      # sorted candidate indices
    dt_hf = timed(lambda: m.eval_forward(batch), n=reps)
    dt = graph_leg(m, batch, out, reps)
    line = (f"C2 BLIP2-Flan-T5-xl + TGB (fusion, flow length {L}, map B), 32 -> {nframe} frames, concat prefix {tuple(st['language_model_inputs'].shape)}, "
            f"T5 greedy 16 tokens: {B / dt:.1f} clips/s at {B} clips per call (T5GreedyDecoder, hipGraph; HF generate, eager: {B / dt_hf:.1f})")
    print(line)
    return dict(clips_per_s=B / dt, line=line)


def sub(cmd, last=1):
    r = subprocess.run([sys.executable] + cmd, cwd=REPO, capture_output=True, text=True)
    return "\n".join((r.stdout.strip().splitlines() or [r.stderr.strip()[-300:]])[-last:])


def main():
    which = [a.lower() for a in sys.argv[1:]] or ["c1", "c2", "c3", "c4", "c5"]
    dev = torch.device("cuda:0")
    with tempfile.TemporaryDirectory() as tmp:
        if "c1" in which:
            c1(dev, tmp)
            torch.cuda.empty_cache()
        if "c2" in which:
            c2(dev, tmp)
            torch.cuda.empty_cache()
    import json
    if "c3" in which:
        d = json.loads(sub(["bench.py", "--no-secondary", "--no-cpu-baseline", "--no-prof", "--steps", "3", "--warmup", "1"]))
        print(f"C3 InstructBLIP-Vicuna-7B + TGB, RAFT inline, T=96 -> 8: {d['value']} clips/s ({d['config']['clips_per_gpu_per_step']} clips per step)")
    if "c4" in which:
        d = json.loads(sub(["bench.py", "--T", "256", "--clips", "48", "--raft-clips", "12", "--no-secondary", "--no-cpu-baseline", "--no-prof", "--steps", "3",
                            "--warmup", "1"]))
        print(f"C4 same, T=256 -> 8 (per GPU; clips shard over the 8 GPUs with no collective): {d['value']} clips/s ({d['config']['clips_per_gpu_per_step']} clips per step)")
    if "c5" in which:
        print("C5 " + sub(["tools/train_bench.py", "4"], last=3))       # the step line, the device-time split, the prefix graph's GEMM rate


if __name__ == "__main__":
    main()
