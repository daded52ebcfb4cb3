"""Do two independent kernels overlap when forked onto a side stream (streams.fork), eagerly and inside a replayed
hipGraph?  Two GEMMs of 2048 x 1280 x 1280 (32 workgroups each on 256 CUs) back to back vs forked."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
from i2v_adapter_unofficial_amd import streams
k = pkg.kernels; dev = torch.device("cuda:0")
M, N, K = int(os.environ.get("M", 2048)), 1280, 1280
a1, a2 = (torch.randn(M, K, device=dev).half() for _ in range(2))
w1, w2 = ((torch.randn(N, K, device=dev) * K ** -0.5).half() for _ in range(2))
o1, o2 = (torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(2))
def serial():
    k.gemm(a1, w1, out=o1); k.gemm(a2, w2, out=o2)
def forked():
    with streams.fork(True, dev) as fk:
        with fk.side():
            k.gemm(a2, w2, out=o2)
        k.gemm(a1, w1, out=o1)
def one():
    k.gemm(a1, w1, out=o1)
def t_eager(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
def t_graph(fn, reps=20):
    fn(); torch.cuda.synchronize(); g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): g.replay()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / (10 * reps) * 1e3
print("env", {k_: v for k_, v in os.environ.items() if "GRAPH" in k_ or "HIP_" in k_})
print(f"M={M} eager: one {t_eager(one):.1f} us  serial pair {t_eager(serial):.1f}  forked pair {t_eager(forked):.1f}")
print(f"M={M} graph: one {t_graph(one):.1f} us  serial pair {t_graph(serial):.1f}  forked pair {t_graph(forked):.1f}")
