"""The step's stride-1 3x3 convolutions at the 64 x 64, 32 x 32 and 16 x 16 levels: time (hipGraph replays over DIFFERENT inputs)
and the max error against torch fp32 on two images."""
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
tag = os.environ.get("AB_TAG", "")
torch.manual_seed(0)
for hw, cin, cout, res in [(64, 320, 320, True), (64, 640, 320, False), (64, 960, 320, False), (32, 640, 640, True),
                           (32, 1280, 640, False), (32, 1920, 640, False), (32, 320, 640, False), (16, 1280, 1280, True),
                           (16, 2560, 1280, False), (16, 640, 1280, False)]:
    n = 6
    xs = [torch.randn(32, hw, hw, cin, device=dev).half() for _ in range(n)]
    wt = torch.randn(cout, cin, 3, 3, device=dev) * (9 * cin) ** -0.5
    w = pack_conv3x3(wt); b = (torch.randn(cout, device=dev) * 0.1).half()
    r = torch.randn(32, hw, hw, cout, device=dev).half() if res else None
    fns = [(lambda x=x: k.conv3x3(x, w, b, residual=r)) for x in xs]
    out = fns[0]()
    ref = torch.nn.functional.conv2d(xs[0][:2].permute(0, 3, 1, 2).float(), wt.half().float(), b.float(), padding=1).permute(0, 2, 3, 1)
    if res: ref = ref + r[:2].float()
    err = (out[:2].float() - ref).abs().max().item() / ref.abs().max().item()
    t = timeit(fns); fl = 2.0 * 32 * hw * hw * cout * 9 * cin
    print(f"[{tag}] conv {32 * hw * hw}x{cout}x{9 * cin} {'+res' if res else '    '} {t:8.1f} us {fl / t * 1e-6:7.1f} TFLOP/s  rel err {err:.2e}", flush=True)
