import sys, gc
sys.path.insert(0, "/root/repo")
import torch
from antsrl_amd import vmm
torch.zeros(1, device="cuda:0")
f0 = torch.cuda.mem_get_info()[0]
for i in range(12):
    t = vmm.pieced_u8(1 << 30, "cuda:0")
    t[:1024].fill_(1)
    fa = torch.cuda.mem_get_info()[0]
    del t
    gc.collect()
    torch.cuda.synchronize()
    print(i, "held while alive %.0f MiB, after free %.0f MiB below start" % ((f0 - fa) / 2**20, (f0 - torch.cuda.mem_get_info()[0]) / 2**20))
