"""Per-kernel timing at the shapes of BASELINE config 2 (16 f x 512^2, CFG batch 2 => 32 images).
Prints achieved TFLOP/s (MFMA-bound kernels) or GB/s (HBM-bound kernels).  Usage: python tools/kernel_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg  # noqa: E402

k = pkg.kernels
dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).half()


def bench_gemm():
    print("== GEMM  M N K  ms  TFLOP/s")
    for M, N, K in [(131072, 320, 320), (131072, 640, 320), (131072, 2560, 320), (131072, 320, 1280),
                    (32768, 640, 640), (32768, 5120, 640), (32768, 640, 2560), (8192, 1280, 1280),
                    (8192, 10240, 1280), (8192, 1280, 5120), (2048, 1280, 1280), (8192, 8192, 8192)]:
        a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        t = timeit(lambda: k.gemm(a, w))
        print(f"gemm {M:7d} {N:6d} {K:6d}  {t * 1e3:8.3f}  {2.0 * M * N * K / t / 1e12:8.1f}")


def bench_conv():
    print("== conv3x3  n h w cin cout  ms  TFLOP/s")
    for n, h, w, ci, co in [(32, 64, 64, 320, 320), (32, 32, 32, 640, 640), (32, 16, 16, 1280, 1280),
                            (32, 8, 8, 1280, 1280), (32, 8, 8, 2560, 1280), (32, 64, 64, 640, 320),
                            (32, 32, 32, 1280, 640)]:
        x, wp = rnd(n, h, w, ci), rnd(co, 9 * ci, scale=(9 * ci) ** -0.5)
        t = timeit(lambda: k.conv3x3(x, wp))
        print(f"conv {n:3d} {h:3d} {w:3d} {ci:5d} {co:5d}  {t * 1e3:8.3f}  {2.0 * n * h * w * 9 * ci * co / t / 1e12:8.1f}")


def bench_attn():
    print("== attention  bq group heads d lq lk  ms  TFLOP/s")
    for bq, grp, hd, d, lq, lk in [(32, 1, 8, 40, 4096, 4096), (32, 16, 8, 40, 4096, 4096), (32, 1, 8, 80, 1024, 1024),
                                   (32, 16, 8, 80, 1024, 1024), (32, 1, 8, 160, 256, 256), (32, 1, 8, 40, 4096, 77),
                                   (32, 1, 8, 64, 4096, 4096), (8, 1, 16, 128, 4096, 4096)]:
        c = hd * d
        q, kk = rnd(bq * lq, c), rnd(bq // grp * lk, c)
        vt = rnd(bq // grp, c, k.pad8(lk))
        t = timeit(lambda: k.attention(q, kk, vt, batch_q=bq, lq=lq, lk=lk, heads=hd, head_dim=d, kv_group=grp))
        print(f"attn {bq:3d} {grp:3d} {hd:2d} {d:4d} {lq:5d} {lk:5d}  {t * 1e3:8.3f}  {4.0 * bq * lq * lk * c / t / 1e12:8.1f}")


def bench_hbm():
    print("== HBM-bound kernels  ms  GB/s (algorithmic bytes)")
    for n, hw, c in [(32, 64, 320), (32, 32, 640), (32, 16, 1280)]:
        x = rnd(n, hw, hw, c)
        g, b = rnd(c), rnd(c)
        t = timeit(lambda: k.groupnorm(x, g, b, 32, 1e-5, silu=True))
        print(f"groupnorm+silu {n}x{hw}x{hw}x{c}  {t * 1e3:8.3f}  {3.0 * x.numel() * 2 / t / 1e9:8.0f}  (2 reads + 1 write)")
        x2 = x.view(-1, c)
        t = timeit(lambda: k.layernorm(x2, g, b, 1e-5))
        print(f"layernorm      {x2.shape[0]}x{c}  {t * 1e3:8.3f}  {2.0 * x.numel() * 2 / t / 1e9:8.0f}")
        npix, F = 2 * hw * hw, 16
        q, kk = rnd(npix * F, c), rnd(npix * F, c)
        vt = rnd(npix, c, 16)
        t = timeit(lambda: k.temporal_attention(q, kk, vt, n_pixels=npix, frames=F, heads=8, head_dim=c // 8))
        print(f"temporal attn  {npix}x{F}x{c}  {t * 1e3:8.3f}  {4.0 * q.numel() * 2 / t / 1e9:8.0f}")


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "conv", "attn", "hbm"]
    t0 = time.time()
    for wname in which:
        globals()[f"bench_{wname}"]()
    print(f"total {time.time() - t0:.1f}s")
