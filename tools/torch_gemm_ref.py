import torch, time
for (T,N,K) in ((2048,28672,8192),(8192,28672,8192),(2048,8192,8192)):
    a = torch.randn(T,K,device='cuda',dtype=torch.float16); b = torch.randn(N,K,device='cuda',dtype=torch.float16)
    for _ in range(3): c = a @ b.t()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): c = a @ b.t()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(f"torch fp16 matmul T={T} N={N} K={K}: {ms*1e3:.0f} us, {2*T*N*K/ms/1e9:.0f} TFLOP/s")
