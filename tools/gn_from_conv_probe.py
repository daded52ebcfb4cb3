"""GroupNorm statistics from the producing convolution's epilogue (i2v_gemm_params.gn_partial) against the norm's own statistics
pass: conv1 (+ time-embedding row) -> GroupNorm + SiLU of a ResnetBlock2D at the 64^2 / 32^2 / 16^2 levels: difference and time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
from i2v_adapter_unofficial_amd.blocks import pack_conv3x3
k = pkg.kernels; dev = torch.device("cuda:0")
def timeit(fns):
    for f in fns: f()
    torch.cuda.synchronize(); g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns: f()
    for _ in range(2): g.replay()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): g.replay()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / (5 * len(fns)) * 1e3
torch.manual_seed(0)
for nimg, hw, cin, cout in [(32, 64, 320, 320), (32, 32, 640, 640), (32, 32, 320, 640), (32, 16, 1280, 1280), (2, 16, 320, 320)]:
    xs = [torch.randn(nimg, hw, hw, cin, device=dev).half() for _ in range(4)]
    wt = torch.randn(cout, cin, 3, 3, device=dev) * (9 * cin) ** -0.5
    w = pack_conv3x3(wt); b = (torch.randn(cout, device=dev) * 0.1 + 0.5).half()
    tv = (torch.randn(2, cout, device=dev) * 2).half()
    ga, be = (1 + 0.1 * torch.randn(cout, device=dev)).half(), (0.1 * torch.randn(cout, device=dev)).half()
    rpv = (nimg // 2) * hw * hw
    def old(x):
        h = k.conv3x3(x, w, b, rowvec=tv, rows_per_vec=rpv)
        return k.groupnorm(h, ga, be, 32, 1e-5, silu=True)
    def new(x):
        h, st = k.conv3x3(x, w, b, rowvec=tv, rows_per_vec=rpv, gn_stats_groups=32)
        return k.groupnorm(h, ga, be, 32, 1e-5, silu=True, stats=st), st
    a = old(xs[0]); bb, st = new(xs[0])
    print(f"conv {nimg * hw * hw}x{cout}x{9 * cin} + GN: partials {'from the epilogue, rows ' + str(st[1]) if st else 'NOT available (own statistics pass)'}; "
          f"max |diff| {(a.float() - bb.float()).abs().max().item():.3e} max |ref| {a.float().abs().max().item():.2f}", flush=True)
    to, tn = timeit([(lambda x=x: old(x)) for x in xs]), timeit([(lambda x=x: new(x)[0]) for x in xs])
    print(f"    statistics pass {to:8.1f} us   from the epilogue {tn:8.1f} us   ({tn - to:+.1f})", flush=True)
