import sys, torch
sys.path.insert(0, "/root/repo")
from r3d_amd import ops
from tools.r02_profile import token_like
x = token_like(128, 128, 7).cuda()
sig, st, af = torch.empty(1,128,device="cuda"), torch.empty(1,4,device="cuda"), torch.empty(1,128,128,device="cuda")
for _ in range(3):
    ops.erank_jacobi(x, sig, st, af_t=af)
torch.cuda.synchronize()
print(st.cpu())
