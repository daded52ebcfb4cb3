"""Time the fused motion-module attention sub-block (i2v_motion_attn_f16) against the launches it replaces at the SD-1.5
64^2 level (131072 rows = CFG 2 x 16 frames x 4096 pixels, C = 320, 8 heads of 40).
With a -DI2V_MA_STAMPS build (bash tools/build_variant.sh ma_stamps "-DI2V_MA_STAMPS"; I2V_LIB_PATH=.ab_libs/ma_stamps.so) and
--stamps: the s_memtime timeline of every wave of the first tiles."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

dev = torch.device("cuda:0")
stamps = "--stamps" in sys.argv
outp = "--outp" in sys.argv          # the out-projection + residual inside the launch (w_o, b_o)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
rows = int(args[0]) if args else 131072
c, heads, d, F = 320, 8, 40, 16
nwg = min(256, rows // 128)
if stamps:       # the kernel reads the buffer's address from the environment (stamps build only)
    sbuf = torch.zeros(nwg * 4 * 8 * 8, dtype=torch.int64, device=dev)
    os.environ["I2V_MA_STAMP_PTR"] = str(sbuf.data_ptr())
import i2v_adapter_unofficial_amd as pkg  # noqa: E402
K = pkg.kernels
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(rows, c, device=dev, generator=g).half()
gamma, beta = torch.ones(c, device=dev).half(), torch.zeros(c, device=dev).half()
pe = torch.randn(32, c, device=dev, generator=g).half()
wq, wk, wv, wo = (torch.randn(c, c, device=dev, generator=g).mul(c ** -0.5).half() for _ in range(4))
w = K.pack_motion_qkv(wq, wk, wv, heads)
g32, s32 = K.motion_attn_tables(gamma, beta, pe, F)
wqk = torch.cat([wq, wk], 0).contiguous()


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
    nl = K.layernorm(x, gamma, beta, 1e-5, pe=pe, pe_period=F)
    qk = K.gemm(nl, wqk)
    vt = K.project_vt(nl, wv, F)
    return K.temporal_attention(qk[:, :c], qk[:, c:], vt, n_pixels=rows // F, frames=F, heads=heads, head_dim=d)


op = K.pack_attn_out(wo, torch.zeros(c, device=dev).half(), heads) if outp else None


def new():
    return K.motion_attn(x, g32, s32, w, heads=heads, head_dim=d, frames=F, eps=1e-5, out_proj=op)


if outp:
    _old = old

    def old():
        return K.gemm(_old(), wo, None, residual=x)


a, b = old(), new()
print("max |fused - unfused|", (a.float() - b.float()).abs().max().item(), "max |ref|", a.float().abs().max().item())
t_old, t_new = timeit(old), timeit(new)
print(f"un-fused (LN, q|k GEMM, V^T GEMM, temporal attention): {t_old:8.1f} us")
print(f"fused i2v_motion_attn_f16:                             {t_new:8.1f} us")
flop = 2.0 * rows * c * 3 * c
print(f"  projections {flop / 1e9:.1f} GFLOP -> {flop / t_new / 1e6:.0f} TFLOP/s; t + o traffic {2 * rows * c * 2 / 1e6:.0f} MB")
if stamps:
    torch.cuda.synchronize()
    st = sbuf.view(nwg, 4, 8, 8).cpu().double()
    if st.abs().sum() == 0:
        print("no stamps: not a -DI2V_MA_STAMPS build")
        sys.exit(0)
    names = ["first LN", "barrier", "pass q", "pass k+softmax", "pass v", "PV+stores", "LN next" + (" + out-proj" if outp else "")]
    tiles = min(4, (rows // 128 + nwg - 1) // nwg)
    for it in range(tiles):
        blk = st[:, it]
        base = blk[:, :, 2].min(dim=1, keepdim=True).values
        print(f"tile {it} of each workgroup: cycles from the tile's first pass start to the end of each phase, mean over workgroups")
        print("          " + " ".join(f"{n:>14s}" for n in names[2:]))
        for wv in range(8):
            print(f"  wave {wv}: " + " ".join(f"{v:14.0f}" for v in (blk[:, wv, 3:8] - base).mean(dim=0).tolist()) +
                  f"   (starts at {(blk[:, wv, 2] - base[:, 0]).mean():.0f})")
    print(f"first LN + barrier of a workgroup: {(st[:, 0, :, 2] - st[:, 0, :, 0]).mean():.0f} cycles")
