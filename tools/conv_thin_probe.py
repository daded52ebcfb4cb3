"""Narrow-output 3x3 convolutions (conv_thin.hip) against torch fp32 and against the tile kernels (I2V_CONV_THIN=0 in a second run):
the UNet's conv_out (320 -> 4, fp32 result) at the 64^2 level and the VAE decoder's (128 -> 3) at 512^2."""
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
print("I2V_CONV_THIN =", os.environ.get("I2V_CONV_THIN", "(default: on)"))
for nimg, hw, cin, cout, f32 in [(32, 64, 320, 4, True), (32, 64, 320, 4, False), (4, 512, 128, 3, False), (2, 24, 64, 16, False), (3, 8, 192, 5, True)]:
    xs = [torch.randn(nimg, hw, hw if hw != 24 else 32, cin, device=dev).half() for _ in range(3)]
    wt = torch.randn(cout, cin, 3, 3, device=dev) * (9 * cin) ** -0.5
    w = pack_conv3x3(wt); b = (torch.randn(cout, device=dev) * 0.1).half()
    fns = [(lambda x=x: k.conv3x3(x, w, b, out_f32=f32)) for x in xs]
    out = fns[0]()
    ref = torch.nn.functional.conv2d(xs[0][:2].permute(0, 3, 1, 2).float(), wt.half().float(), b.float(), padding=1).permute(0, 2, 3, 1)
    err = (out[:2].float() - ref).abs().max().item() / ref.abs().max().item()
    t = timeit(fns)
    mb = xs[0].numel() * 2 / 1e6
    print(f"conv {tuple(xs[0].shape)} -> {cout} {'fp32' if f32 else 'fp16'}: {t:8.1f} us  input {mb:.0f} MB -> {mb / t * 1e-3 * 1e3:.2f} TB/s of input  rel err {err:.2e}", flush=True)
