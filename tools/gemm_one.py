import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from curla_amd import ops
B,H=512,1024
x, W, bias = torch.randn(B,H,device="cuda"), torch.randn(H,H,device="cuda"), torch.randn(H,device="cuda")
out = torch.empty(B,H,device="cuda")
for _ in range(20):
    ops.linear_fwd(x,0,W,0,bias,0,out,0,B,H,H,1,relu=1)
torch.cuda.synchronize()
