import torch, time
dev="cuda:0"; M=257*248
g=torch.Generator(device=dev).manual_seed(0)
for name,n,k in (("qkv",4224,1408),("proj",1408,1408),("fc1",6144,1408),("fc2",1408,6144)):
    A=torch.randn(M,k,generator=g,device=dev).bfloat16(); W=(torch.randn(n,k,generator=g,device=dev)*0.05).bfloat16(); b=torch.randn(n,device=dev).bfloat16()
    for _ in range(3): torch.nn.functional.linear(A,W,b)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): torch.nn.functional.linear(A,W,b)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(f"hipBLASLt {name}: {ms:.3f} ms {2.0*M*n*k/ms/1e9:.1f} TF/s (bf16 out, bias, no gelu/residual)")
