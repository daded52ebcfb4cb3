import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
bq, grp, hd, d, lq, lk = 32, int(os.environ.get("GRP", "16")), 8, int(os.environ.get("D", "40")), 4096, 4096
c = hd * d
q = torch.randn(bq * lq, c, device=dev).half(); kk = torch.randn(bq // grp * lk, c, device=dev).half()
vt = torch.randn(bq // grp, c, lk, device=dev).half()
for _ in range(3):
    k.attention(q, kk, vt, batch_q=bq, lq=lq, lk=lk, heads=hd, head_dim=d, kv_group=grp)
torch.cuda.synchronize()
