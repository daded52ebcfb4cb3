"""The 64^2-level 320 -> 320 convolution with its three epilogues (plain, + time-embedding row vector per image, + residual), B = 32
and B = 16: does the row-vector form cost more than the residual form?  (r5_step_shapes: 271 us against 224 us.)"""
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
hw, cin, cout = 64, 320, 320
wt = torch.randn(cout, cin, 3, 3, device=dev) * (9 * cin) ** -0.5
w = pack_conv3x3(wt); b = (torch.randn(cout, device=dev) * 0.1).half()
for nimg in (32, 16):
    xs = [torch.randn(nimg, hw, hw, cin, device=dev).half() for _ in range(6)]
    r = torch.randn(nimg, hw, hw, cout, device=dev).half()
    tv = torch.randn(nimg, cout, device=dev).half()
    for name, kw in (("plain", {}), ("+ rowvec per image", dict(rowvec=tv, rows_per_vec=hw * hw)), ("+ residual", dict(residual=r))):
        t = timeit([(lambda x=x: k.conv3x3(x, w, b, **kw)) for x in xs])
        fl = 2.0 * nimg * hw * hw * cout * 9 * cin
        print(f"conv {nimg * hw * hw}x{cout}x{9 * cin} {name:20s} {t:8.1f} us {fl / t * 1e-6:7.1f} TFLOP/s", flush=True)
