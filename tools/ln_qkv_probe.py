"""Time the one-launch LayerNorm + [q | k | q_adapter] + V^T projection (i2v_ln_qkv_f16) against the two LayerNorm-folded GEMMs
it replaces at the SD-1.5 64^2 level (131072 rows = 32 images of 4096 tokens, C = 320), on HBM-cold operands (six activation
sets in turn, as inside a denoising step)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import i2v_adapter_unofficial_amd as pkg  # noqa: E402
from i2v_adapter_unofficial_amd.blocks import fold_layernorm  # noqa: E402

K = pkg.kernels
dev = torch.device("cuda:0")
n_img, L, c = 32, 4096, 320
rows = n_img * L
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn(rows, c, device=dev, generator=g).half() for _ in range(6)]
gamma, beta = (1 + 0.1 * torch.randn(c, device=dev, generator=g)).half(), (0.1 * torch.randn(c, device=dev, generator=g)).half()
w_qk = torch.randn(3 * c, c, device=dev, generator=g).mul(c ** -0.5).half()
w_v = torch.randn(c, c, device=dev, generator=g).mul(c ** -0.5).half()
wp = K.pack_ln_qkv(w_qk, w_v)
f_qk, f_v = fold_layernorm(w_qk, None, gamma, beta), fold_layernorm(w_v, None, gamma, beta)
g32, b32 = gamma.float(), beta.float()
it = [0]


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
    return e0.elapsed_time(e1) / n * 1e3


def old():
    x = xs[it[0] % 6]
    it[0] += 1
    qk = K.gemm(x, f_qk[0], f_qk[2], ln=(f_qk[1], 1e-5))
    return qk, K.project_vt(x, f_v[0], L, bias=f_v[2], ln=(f_v[1], 1e-5))


def new():
    x = xs[it[0] % 6]
    it[0] += 1
    return K.ln_qkv(x, g32, b32, wp, n_qk=3 * c, rows_per_image=L, eps=1e-5)


it[0] = 0
a = old()
it[0] = 0
b = new()
print("max |fused - unfused| q|k|q'", (a[0].float() - b[0].float()).abs().max().item(), " V^T", (a[1].float() - b[1].float()).abs().max().item())
flop = 2.0 * rows * c * 4 * c
t_old, t_new = timeit(old), timeit(new)
print(f"LayerNorm-folded q|k|q' GEMM + V^T GEMM: {t_old:8.1f} us ({flop / t_old / 1e6:.0f} TFLOP/s)")
print(f"i2v_ln_qkv_f16:                          {t_new:8.1f} us ({flop / t_new / 1e6:.0f} TFLOP/s)")
