"""One attention shape in isolation (for rocprofv3 --pmc passes and env-override sweeps): prints ms and TFLOP/s.
env: D (head_dim, 40), GRP (kv_group, 16), LQ / LK (4096), HEADS (8), BQ (32); I2V_ATTN_QT / I2V_ATTN_KVT are read by the
library itself."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
E = lambda n, v: int(os.environ.get(n, v))
bq, grp, hd, d, lq, lk = E("BQ", 32), E("GRP", 16), E("HEADS", 8), E("D", 40), E("LQ", 4096), E("LK", 4096)
c = hd * d
q = torch.randn(bq * lq, c, device=dev).half(); kk = torch.randn(bq // grp * lk, c, device=dev).half()
vt = torch.randn(bq // grp, c, (lk + 7) // 8 * 8, device=dev).half()
run = lambda: k.attention(q, kk, vt, batch_q=bq, lq=lq, lk=lk, heads=hd, head_dim=d, kv_group=grp)
for _ in range(3):
    run()
torch.cuda.synchronize()
n = E("ITERS", 20)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"attn d={d} grp={grp} lq={lq} lk={lk} QT={os.environ.get('I2V_ATTN_QT', '-')} KVT={os.environ.get('I2V_ATTN_KVT', '-')}"
      f"  {ms:.3f} ms  {4.0 * bq * hd * lq * lk * d / ms / 1e9:.1f} TFLOP/s")
