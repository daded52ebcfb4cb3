"""Weight gradient dW = dY^T X of the adapter projections (M = N = C, contraction over all tokens) at the UNet's levels."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
from i2v_adapter_unofficial_amd.training import wgrad
K = pkg.kernels; dev = torch.device("cuda:0")
torch.manual_seed(0)
for tokens, c in ((65536, 320), (16384, 640), (4096, 1280), (1024, 1280)):
    dy = torch.randn(tokens, c, device=dev).half(); x = torch.randn(tokens, c, device=dev).half()
    ref = (dy.float().T @ x.float())
    got = wgrad(dy, x)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dyt = K.transpose_tokens(dy, tokens).view(c, -1); xt = K.transpose_tokens(x, tokens).view(c, -1)
    e0.record()
    for _ in range(10): K.gemm(dyt, xt, out_scale=2.0 ** -6)
    e1.record(); torch.cuda.synchronize(); g = e0.elapsed_time(e1) / 10
    e0.record()
    for _ in range(10): wgrad(dy, x)
    e1.record(); torch.cuda.synchronize(); w = e0.elapsed_time(e1) / 10
    print(f"wgrad tokens {tokens} C {c}: {w * 1e3:7.1f} us in all, GEMM {c}x{c}x{tokens} alone {g * 1e3:7.1f} us ({2.0 * c * c * tokens / g / 1e9:6.1f} TFLOP/s), rel err {err:.1e}")
