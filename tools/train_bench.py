#!/usr/bin/env python3
"""Config C5 micro-step at full size (row a14): InstructBLIP-Vicuna-7B geometry, the REFERENCE's trainable set -- Q-Former
(185.7 M) + query_tokens + language_projection + temporal_projection + LoRA r=8 on q_proj / v_proj (4.19 M) -- per micro-batch:
frozen EVA-ViT-g over the B x 8 pre-selected frames (HIP) -> Q-Former + mean pool + projection (HIP forward, PyTorch-recompute
backward) -> [32 prefix | 48 question | 31 answer tokens] through the LLM with LoRA -> HIP token splice / labels / shifted CE
-> backward; AdamW + one flat-bucket gradient all-reduce every 4 micro-batches.
    python tools/train_bench.py [B] [--lora-only]      prints ms per micro-batch and sequences/s on one MI355X.
Under torchrun (2+ ranks, backend nccl = RCCL) the flat bucket is all-reduced across the ranks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from videotgb_amd import llm, models, synth, train
world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if args else 4                            # per-GPU micro-batch of the reference experiment config
lora_only = "--lora-only" in sys.argv
cfg = synth.full_cfg("instructblip")
lm = llm.build_llama("vicuna-7b", torch.bfloat16, dev)
m = models.LSTP(cfg, dev, language_model=lm, compute_dtype="bf16", raft_dtype="bf16")
sd = synth.path_state_dict(cfg, 0, with_raft=False)
m.load_state_dict(sd, strict=False); m.to(dev); lm.to(torch.bfloat16)
step = train.LoraTrainStep(m, pad_token_id=0, lr=1e-4, accumulate_grad_batches=4, train_prefix=not lora_only)
lm.train()
n_train = sum(p.numel() for p in step.params)
g = torch.Generator(device=dev).manual_seed(local)
P, Li, Lo, nframe = 32, 48, 32, 8
frames = torch.randn(B * nframe, 3, 224, 224, generator=g, device=dev)
qt = torch.randint(1000, 30000, (B, 14), generator=g, device=dev); qtm = torch.ones_like(qt)
q = torch.randint(3, 32000, (B, Li), generator=g, device=dev); qm = torch.ones_like(q)
a = torch.randint(3, 32000, (B, Lo), generator=g, device=dev); am = torch.ones_like(a)
def micro():
    return step.step_frames(frames, qt, qtm, [nframe] * B, q, qm, a, am)
for _ in range(4): loss, _ = micro()
torch.cuda.synchronize(); t0 = time.time()
n = 8
for _ in range(n): loss, stepped = micro()
torch.cuda.synchronize(); dt = (time.time() - t0) / n
if local == 0:
    print(f"C5 micro-batch B={B} x {world} GPU(s), S={P + Li + Lo - 1}, trainable {n_train} params ({n_train * 4 / 1e6:.0f} MB fp32 gradient bucket"
          f"{', adapters only' if lora_only else ''}): {dt * 1e3:.1f} ms per micro-batch incl. the prefix path ({B * world / dt:.1f} sequences/s), "
          f"loss {loss.item():.3f}, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
if local == 0 and "--no-profile" not in sys.argv:
    # where one micro-batch's device time goes, and the rate of the prefix graph's own GEMMs (forward + dgrad + wgrad, operands in place)
    from torch.profiler import ProfilerActivity, profile
    train.GEMM_FLOPS[0] = 0.0
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        micro(); torch.cuda.synchronize()
    fam = {"own training GEMM (tg_*)": 0.0, "own attention / LayerNorm / GELU / CE (training)": 0.0, "own inference kernels (frozen ViT-g ...)": 0.0,
           "torch: LLM GEMMs (hipBLASLt)": 0.0, "torch: other": 0.0}
    for e in prof.key_averages():
        t, k = e.device_time_total / 1e3, e.key
        if t <= 0: continue
        if "tg_mfma_kernel" in k or "tg_f32_kernel" in k or "tg_split_reduce" in k: fam["own training GEMM (tg_*)"] += t
        elif any(x in k for x in ("attn_train", "ln_train", "gelu_fwd", "gelu_bwd", "col_sum", "shifted_ce", "concat_text_io")): fam["own attention / LayerNorm / GELU / CE (training)"] += t
        elif any(x in k for x in ("gemm_bf16", "attn_bf16", "layernorm_kernel", "gemm_skinny", "vit_", "pool_", "qformer")): fam["own inference kernels (frozen ViT-g ...)"] += t
        elif "Cijk_" in k: fam["torch: LLM GEMMs (hipBLASLt)"] += t
        else: fam["torch: other"] += t
    if "--kernels" in sys.argv:
        for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:30]:
            print(f"      {e.device_time_total / 1e3:8.2f} ms x{e.count:5d}  {e.key[:150]}")
    tot = sum(fam.values())
    print("   device time of one micro-batch: " + "; ".join(f"{k} {v:.1f} ms ({100 * v / tot:.0f} %)" for k, v in fam.items()))
    tg = fam["own training GEMM (tg_*)"]
    if tg > 0:
        tf = train.GEMM_FLOPS[0] / (tg * 1e-3) / 1e12
        print(f"   prefix-graph GEMMs: {train.GEMM_FLOPS[0] / 1e12:.2f} TFLOP in {tg:.1f} ms = {tf:.0f} TFLOP/s = {100 * tf / 2500:.1f} % of the 2.5 PFLOP/s dense bf16 peak")
if world > 1:
    dist.barrier(); dist.destroy_process_group()
