import sys, torch; sys.path.insert(0, ".")
from exllama_amd import cuda_ext as ce
for M,K,N in [(1,4096,16),(2048,4096,16),(2048,16,4096),(2048,64,11008),(1,4096,64)]:
    x=torch.randn(M,K,device="cuda:0").half(); w=(torch.randn(K,N,device="cuda:0")*0.1).half()
    for _ in range(3): ce.ext_half_matmul(x,w,cublas=True)
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ce.ext_half_matmul(x,w,cublas=True)
    e1.record(); torch.cuda.synchronize()
    print(M,K,N, f"{e0.elapsed_time(e1)*1e3/20:.1f} us")
