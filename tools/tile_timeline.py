"""Timeline of one output tile of the 8-wave GEMM from s_memrealtime stamps written by a -DI2V_PROBE=5 build
(tools/build_variant.sh probe5 "-DI2V_PROBE=5" gemm_big.hip; run with I2V_LIB_PATH=.ab_libs/probe5.so)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
for N, K, epi, M, res in ((320, 320, 0, 131072, False), (320, 320, 0, 131072, True), (320, 1280, 0, 131072, True),
                          (2560, 320, k.I2V_EPI_GEGLU, 131072, False), (640, 640, 0, 32768, False), (5120, 640, k.I2V_EPI_GEGLU, 32768, False)):
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half(); b = torch.randn(N, device=dev).half()
    a = torch.randn(M, K, device=dev).half()
    r = torch.randn(M, N, device=dev).half() if res else None
    No = N // 2 if epi else N
    out = torch.empty(M, No, device=dev, dtype=torch.float16)
    flush = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    for _ in range(3):
        flush.zero_()
        k.gemm(a, w, b, out=out, epilogue=epi, residual=r)
    torch.cuda.synchronize()
    tn = N // 320
    oc = 160 if epi else 320
    rows = out[::256].contiguous().view(M // 256, No)
    raw = torch.stack([rows[:, j * oc: j * oc + 48].contiguous().view(torch.int64) for j in range(tn)], 1).reshape(-1, 12).cpu().double()
    st = raw[:, :6] / 100.0  # us
    ck = raw[:, 6:]
    ghz = (ck[:, 1:] - ck[:, :-1]) / (st[:, 1:] - st[:, :-1]).clamp(min=0.01) / 1e3
    t0 = st[:, 0].min()
    d = st[:, 1:] - st[:, :-1]
    span = st[:, 5].max() - t0
    names = ["prologue+1st tile", "K loop", "barrier/LN/bias", "stage+issue stores", "store drain"]
    print(f"N={N} K={K} epi={epi} res={res} M={M}: {st.shape[0]} tiles, kernel span {span:.1f} us, tile total mean {(st[:,5]-st[:,0]).mean():.2f} us")
    print("   " + "  ".join(f"{n}: {d[:, i].mean():.2f} (p90 {d[:, i].quantile(0.9):.2f}) @{ghz[:, i].median():.2f} GHz" for i, n in enumerate(names)))
    # tile rounds: start times cluster; report the mean start of each quartile of tiles
    s0 = (st[:, 0] - t0).sort().values
    nt = st.shape[0]
    print("   start of tiles (us since first): " + " ".join(f"{s0[int(q * (nt - 1))]:.1f}" for q in (0, 0.1, 0.25, 0.4, 0.5, 0.6, 0.75, 0.9, 1.0)))
