"""Attention backward at the 64 x 64 level of the training step (16 frames x 4096 tokens, 8 heads x 40): self-attention
(kv_group 1) and the cross-frame form (kv_group 16: dK0 / dV0 summed over the frames), per-kernel times from torch events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
K = pkg.kernels; dev = torch.device("cuda:0")
torch.manual_seed(0)
bq, heads, d, L = int(os.environ.get("BQ", 16)), 8, int(os.environ.get("D", 40)), int(os.environ.get("L", 4096))
C = heads * d
for group in (1, bq):
    q = torch.randn(bq * L, C, device=dev).half(); k = torch.randn(bq // group * L, C, device=dev).half()
    v = torch.randn(bq // group * L, C, device=dev).half(); do = torch.randn(bq * L, C, device=dev).half()
    o, lse = K.attention(q, k, K.transpose_tokens(v, L), batch_q=bq, lq=L, lk=L, heads=heads, head_dim=d, kv_group=group, return_lse=True)
    run = lambda: K.attention_bwd(q, k, v, o, do, batch_q=bq, lq=L, lk=L, heads=heads, head_dim=d, kv_group=group, lse=lse)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    print(f"attention_bwd B{bq} L{L} d{d} kv_group {group}: {e0.elapsed_time(e1) / 5:.3f} ms per call (dq + dkv sweeps, transposes, delta)")
