"""The 77-token text cross-attention of the 64 x 64 level (B32 Lq4096 h8 d40) under the current environment, next to the SAME
arithmetic on a head-major layout (heads = 1, batch = 256: every head's q / o rows contiguous 80-byte rows) and to a plain
copy of Q -> O: is the launch bound by the 80-byte head slices of 640-byte token rows?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg  # noqa: E402

k = pkg.kernels
dev = torch.device("cuda:0")
tag = " ".join(f"{e}={os.environ[e]}" for e in ("I2V_ATTN_WALK", "I2V_ATTN_KVT") if e in os.environ) or "default"


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


g = torch.Generator().manual_seed(0)
for (B, L, H, D, LK) in ((32, 4096, 8, 40, 77), (32, 1024, 8, 80, 77), (32, 256, 8, 160, 77), (32, 4096, 8, 40, 4)):
    C = H * D
    q = torch.randn(B * L, C, generator=g).half().to(dev)
    kk = torch.randn(2 * LK, C, generator=g).half().to(dev)
    vt = torch.randn(2, C, k.pad8(LK), generator=g).half().to(dev)
    o = torch.empty_like(q)
    us = timeit(lambda: k.attention(q, kk, vt, batch_q=B, lq=L, lk=LK, heads=H, head_dim=D, kv_group=B // 2, out=o))
    # head-major: one head per "batch" entry, rows of D halves
    qh = torch.randn(B * H * L, D, generator=g).half().to(dev)
    kh = torch.randn(2 * H * LK, D, generator=g).half().to(dev)
    vth = torch.randn(2 * H, D, k.pad8(LK), generator=g).half().to(dev)
    oh = torch.empty_like(qh)
    us_h = timeit(lambda: k.attention(qh, kh, vth, batch_q=B * H, lq=L, lk=LK, heads=1, head_dim=D, kv_group=B // 2, out=oh))
    us_c = timeit(lambda: o.copy_(q))
    mb = 2 * q.numel() * 2 / 1e6
    print(f"{tag}: B{B} Lq{L} Lk{LK} h{H} d{D}: token-major {us:7.1f} us ({mb / us * 1e3:6.0f} GB/s)   head-major {us_h:7.1f} us"
          f"   copy Q->O {us_c:6.1f} us", flush=True)
