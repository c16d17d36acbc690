import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from curla_amd import ops
def timeit(fn, iters=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B,H=512,1024
x, W = torch.randn(B,H,device="cuda"), torch.randn(H,H,device="cuda")
for ks in (1,2,4,8):
    out = torch.empty(ks,B,H,device="cuda")
    us = timeit(lambda: ops.gemm(x,0,H,0,W,0,H,0,out,H,0,B,H,H,1,ksplit=ks,split_stride=B*H))
    print(f"NT 512x1024x1024 ksplit {ks}: {us:.1f} us  {2*B*H*H/us/1e6:.1f} TF")
for M in (512, 1024, 2048, 4096):
    x = torch.randn(M,H,device="cuda"); out = torch.empty(M,H,device="cuda")
    us = timeit(lambda: ops.gemm(x,0,H,0,W,0,H,0,out,H,0,M,H,H,1))
    print(f"NT {M}x1024x1024: {us:.1f} us  {2*M*H*H/us/1e6:.1f} TF")
