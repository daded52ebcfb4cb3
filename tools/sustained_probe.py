"""Is a kernel slower when the chip has been busy for milliseconds (power / clock management) than after an idle gap?
The same launch timed (a) 40x back to back, (b) each launch after ~1.5 ms of idle GPU, (c) back to back again."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
def ev(): return torch.cuda.Event(enable_timing=True)
def sustained(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s, e = ev(), ev(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
def spaced(fn, n=15):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); time.sleep(0.002)
        s, e = ev(), ev(); s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
M = 131072
a = torch.randn(M, 320, device=dev).half(); w = (torch.randn(2560, 320, device=dev) * 320 ** -0.5).half(); b = torch.randn(2560, device=dev).half()
q = torch.randn(32 * 4096, 320, device=dev).half(); kk = torch.randn(32 * 4096, 320, device=dev).half()
vt = torch.randn(32, 320, 4096, device=dev).half()
x = torch.randn(32, 64, 64, 320, device=dev).half(); wc = (torch.randn(320, 2880, device=dev) * 0.02).half()
cases = {"geglu 131072x2560x320": lambda: k.gemm(a, w, b, epilogue=k.I2V_EPI_GEGLU),
         "attention L0 self": lambda: k.attention(q, kk, vt, batch_q=32, lq=4096, lk=4096, heads=8, head_dim=40, kv_group=1, scale=0.158),
         "conv3x3 131072x320x2880": lambda: k.conv3x3(x, wc)}
for name, fn in cases.items():
    s1 = sustained(fn); sp = spaced(fn); s2 = sustained(fn)
    print(f"{name:28s} back-to-back {s1:8.1f} us   after idle {sp:8.1f} us   back-to-back again {s2:8.1f} us")
