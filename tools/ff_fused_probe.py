"""Time the one-launch GEGLU feed-forward (i2v_ff_fused_f16) against the pair it replaces at the SD-1.5 64^2 level
(131072 rows, C = 320, inner 1280): LayerNorm-folded GEGLU GEMM + the output GEMM with the residual."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
stamps = "--stamps" in sys.argv
if stamps:       # a -DI2V_FF_STAMPS build reads the buffer's address from the environment
    sbuf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda:0")
    os.environ["I2V_FF_STAMP_PTR"] = str(sbuf.data_ptr())
import i2v_adapter_unofficial_amd as pkg  # noqa: E402
from i2v_adapter_unofficial_amd.blocks import fold_layernorm_geglu  # noqa: E402

K = pkg.kernels
dev = torch.device("cuda:0")
rows, c, inner = 131072, 320, 1280
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(rows, c, device=dev, generator=g).half()
gamma, beta = (1 + 0.1 * torch.randn(c, device=dev, generator=g)).half(), (0.1 * torch.randn(c, device=dev, generator=g)).half()
w1 = torch.randn(2 * inner, c, device=dev, generator=g).mul(c ** -0.5).half()
b1 = torch.randn(2 * inner, device=dev, generator=g).mul(0.1).half()
w2 = torch.randn(c, inner, device=dev, generator=g).mul(inner ** -0.5).half()
b2 = torch.randn(c, device=dev, generator=g).mul(0.1).half()
packed = K.pack_ff_fused(w1, b1, w2, b2)
wf, ws, bf = fold_layernorm_geglu(w1, b1, gamma, beta)
g32, b32 = gamma.float(), beta.float()


def timeit(fn, n=10):
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
    hmid = K.gemm(x, wf, bf, epilogue=K.I2V_EPI_GEGLU, ln=(ws, 1e-5))
    return K.gemm(hmid, w2, b2, residual=x)


def new():
    return K.ff_fused(x, g32, b32, packed, eps=1e-5)


a, b = old(), new()
print("max |fused - unfused|", (a.float() - b.float()).abs().max().item(), "max |ref|", a.float().abs().max().item())
t_old, t_new = timeit(old), timeit(new)
flop = 2.0 * rows * c * 2 * inner + 2.0 * rows * inner * c
print(f"un-fused (LayerNorm-folded GEGLU GEMM, output GEMM + residual): {t_old:8.1f} us ({flop / t_old / 1e6:.0f} TFLOP/s)")
print(f"fused i2v_ff_fused_f16:                                         {t_new:8.1f} us ({flop / t_new / 1e6:.0f} TFLOP/s)")

# (r5) the Linear that follows the block as the tail of the same launch, against fused feed-forward + i2v_gemm_f16
if hasattr(K, "pack_ff_tail"):
    w3 = torch.randn(c, c, device=dev, generator=g).mul(c ** -0.5).half()
    b3 = torch.randn(c, device=dev, generator=g).mul(0.1).half()
    res2 = torch.randn(rows, c, device=dev, generator=g).half()
    ptail = K.pack_ff_tail(w3, b3)
    for frames, hw in ((0, 0), (16, 4096)):
        store = dict(store=K.I2V_STORE_ROWPERM, frames=frames, hw=hw) if frames else {}
        two = lambda: K.gemm(K.ff_fused(x, g32, b32, packed, eps=1e-5), w3, b3, residual=res2, **store)
        one = lambda: K.ff_fused(x, g32, b32, packed, eps=1e-5, tail=(ptail, res2, frames, hw))
        a2, b2_ = two(), one()
        print(f"tail (frames {frames}): max |one launch - two launches| {(a2.float() - b2_.float()).abs().max().item():.3e}; "
              f"two launches {timeit(two):8.1f} us, one {timeit(one):8.1f} us")

# HBM-cold operands, as inside a denoising step (the 84 MB of x / res2 of the previous launch of this shape left MALL long ago):
# six sets of activations, used in turn
xs = [torch.randn(rows, c, device=dev, generator=g).half() for _ in range(6)]
rs = [torch.randn(rows, c, device=dev, generator=g).half() for _ in range(6)]
os_ = [torch.empty(rows, c, device=dev, dtype=torch.float16) for _ in range(6)]
it = [0]


def cold(fn):
    def run():
        i = it[0] = (it[0] + 1) % 6
        return fn(xs[i], rs[i], os_[i])
    return run


print("HBM-cold operands (six activation sets in turn):")
print(f"  fused feed-forward                      {timeit(cold(lambda a, r, o: K.ff_fused(a, g32, b32, packed, eps=1e-5, out=o)), 30):8.1f} us")
if hasattr(K, "pack_ff_tail"):
    for frames, hw in ((0, 0), (16, 4096)):
        store = dict(store=K.I2V_STORE_ROWPERM, frames=frames, hw=hw) if frames else {}
        t2 = timeit(cold(lambda a, r, o: K.gemm(K.ff_fused(a, g32, b32, packed, eps=1e-5, out=o), w3, b3, residual=r, **store)), 30)
        t1 = timeit(cold(lambda a, r, o: K.ff_fused(a, g32, b32, packed, eps=1e-5, out=o, tail=(ptail, r, frames, hw))), 30)
        print(f"  + proj_out (frames {frames:2d}): two launches {t2:8.1f} us, one {t1:8.1f} us")

if stamps:
    names = ["load + LN", "FF1", "GEGLU + write", "barrier", "FF2", "epilogue (rest)", "wait + barrier", "in place + P pass"]
    runs = [("no tail", new)]
    if hasattr(K, "pack_ff_tail"):
        runs += [("tail", lambda: K.ff_fused(x, g32, b32, packed, eps=1e-5, tail=(ptail, res2, 0, 0))),
                 ("tail, rows permuted", lambda: K.ff_fused(x, g32, b32, packed, eps=1e-5, tail=(ptail, res2, 16, 4096)))]
    for label, fn in runs:
        sbuf.zero_()
        fn()
        torch.cuda.synchronize()
        st = sbuf.view(256, 8, 8).cpu().double()
        print(f"{label}: cycles per workgroup (4 tiles), by phase, mean over workgroups; waves 0-3 / 4-7")
        for i, n in enumerate(names):
            print(f"  {n:18s} {st[:, :4, i].mean():10.0f} {st[:, 4:, i].mean():10.0f}")
        print(f"  total              {st[:, :4].sum(-1).mean():10.0f} {st[:, 4:].sum(-1).mean():10.0f}")
