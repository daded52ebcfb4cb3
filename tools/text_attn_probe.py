"""Time the fused LayerNorm + to_q + text cross-attention (i2v_cross_attn_fused_f16) against the launches it replaces at the
SD-1.5 64^2 level (131072 rows = CFG 2 x 16 frames x 4096 pixels, C = 320, 8 heads of 40, 77 context tokens per prompt)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
stamps = "--stamps" in sys.argv
if stamps:       # a -DI2V_MA_STAMPS build reads the buffer's address from the environment
    sbuf = torch.zeros(256 * 4 * 8 * 8, dtype=torch.int64, device="cuda:0")
    os.environ["I2V_MA_STAMP_PTR"] = str(sbuf.data_ptr())
import i2v_adapter_unofficial_amd as pkg  # noqa: E402
from i2v_adapter_unofficial_amd.blocks import fold_layernorm  # noqa: E402

K = pkg.kernels
dev = torch.device("cuda:0")
rows, c, heads, d, lt, n_ctx = 131072, 320, 8, 40, 77, 2
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(rows, c, device=dev, generator=g).half()
gamma, beta = (1 + 0.1 * torch.randn(c, device=dev, generator=g)).half(), (0.1 * torch.randn(c, device=dev, generator=g)).half()
wq = torch.randn(c, c, device=dev, generator=g).mul(c ** -0.5).half()
k = torch.randn(n_ctx * lt, c, device=dev, generator=g).half()
vt = torch.zeros(n_ctx, c, 80, device=dev).half()
vt[:, :, :lt] = torch.randn(n_ctx, c, lt, device=dev, generator=g).half()
w = K.pack_cross_q(wq, heads)
frag = K.pack_ctx_fragments(k, vt, heads, lt)
wf, ws, cb = fold_layernorm(wq, None, gamma, beta)
g32, b32 = gamma.float(), beta.float()


def timeit(fn, n=20):
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
    q = K.gemm(x, wf, cb, ln=(ws, 1e-5))
    return K.attention(q, k, vt, batch_q=32, lq=4096, lk=lt, heads=heads, head_dim=d, kv_group=16)


def new():
    return K.cross_attn_fused(x, g32, b32, w, frag, heads=heads, head_dim=d, ctx_len=lt, rows_per_ctx=rows // n_ctx, eps=1e-5)


a, b = old(), new()
print("max |fused - unfused|", (a.float() - b.float()).abs().max().item(), "max |ref|", a.float().abs().max().item())
print(f"un-fused (LayerNorm-folded q GEMM, attention Lk = 77): {timeit(old):8.1f} us")
print(f"fused i2v_cross_attn_fused_f16:                        {timeit(new):8.1f} us")

if stamps:
    sbuf.zero_()
    new()
    torch.cuda.synchronize()
    st = sbuf.view(256, 4, 8, 8).cpu().double()
    names = ["q pass", "ctx frags", "-", "attention+stores", "LN next"]
    for it in (1, 2):
        blk = st[:, it]
        base = blk[:, :, 2].min(dim=1, keepdim=True).values
        print(f"tile {it}: cycles from the tile's start to the end of each phase, mean over workgroups:", names)
        for wv in range(8):
            print(f"  wave {wv}: " + " ".join(f"{v:9.0f}" for v in (blk[:, wv, 3:8] - base).mean(dim=0).tolist()))
