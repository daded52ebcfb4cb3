"""The out-projection inside the fused attention kernels (i2v_motion_attn_f16 / i2v_cross_attn_fused_f16 with w_o, b_o) against the
pair of launches it replaces (fused attention, then i2v_gemm_f16 + residual) at the SD-1.5 64^2 level: difference and time."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

dev = torch.device("cuda:0")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
rows = int(args[0]) if args else 131072
frames = int(args[1]) if len(args) > 1 else 16
c, heads, d = 320, 8, 40
import i2v_adapter_unofficial_amd as pkg  # noqa: E402
K = pkg.kernels
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(rows, c, device=dev, generator=g).half()
gamma = (1 + 0.1 * torch.randn(c, device=dev, generator=g)).half()
beta = (0.1 * torch.randn(c, device=dev, generator=g)).half()
pe = torch.randn(32, c, device=dev, generator=g).half()
wq, wk, wv, wo = (torch.randn(c, c, device=dev, generator=g).mul(c ** -0.5).half() for _ in range(4))
bo = (0.1 * torch.randn(c, device=dev, generator=g)).half()
w = K.pack_motion_qkv(wq, wk, wv, heads)
g32, s32 = K.motion_attn_tables(gamma, beta, pe, frames)
op = K.pack_attn_out(wo, bo, heads)


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


def report(name, pair, one):
    a, b = pair(), one()
    ref = a.float()
    print(f"{name}: max |one launch - pair| {(ref - b.float()).abs().max().item():.3e}  max |ref| {ref.abs().max().item():.2f}  "
          f"finite {bool(torch.isfinite(b).all())}")
    tp, to = timeit(pair), timeit(one)
    print(f"  pair {tp:7.1f} us   one launch {to:7.1f} us   ({to - tp:+.1f})")


report(f"motion F={frames}",
       lambda: K.gemm(K.motion_attn(x, g32, s32, w, heads=heads, head_dim=d, frames=frames, eps=1e-5), wo, bo, residual=x),
       lambda: K.motion_attn(x, g32, s32, w, heads=heads, head_dim=d, frames=frames, eps=1e-5, out_proj=op))
print(f"  attention alone {timeit(lambda: K.motion_attn(x, g32, s32, w, heads=heads, head_dim=d, frames=frames, eps=1e-5)):7.1f} us")

# text cross-attention: 77 tokens, two contexts (CFG)
lt, n_ctx = 77, 2
ck = torch.randn(n_ctx * lt, c, device=dev, generator=g).half()
cvt = torch.randn(n_ctx, c, 80, device=dev, generator=g).half()
frag = K.pack_ctx_fragments(ck, cvt, heads, lt)
wqf = K.pack_cross_q(wq, heads)
g2, b2 = gamma.float().contiguous(), beta.float().contiguous()
kw = dict(heads=heads, head_dim=d, ctx_len=lt, rows_per_ctx=rows // n_ctx, eps=1e-5)
report("cross Lk=77",
       lambda: K.gemm(K.cross_attn_fused(x, g2, b2, wqf, frag, **kw), wo, bo, residual=x),
       lambda: K.cross_attn_fused(x, g2, b2, wqf, frag, out_proj=op, **kw))
print(f"  attention alone {timeit(lambda: K.cross_attn_fused(x, g2, b2, wqf, frag, **kw)):7.1f} us")
