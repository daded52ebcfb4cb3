import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
M, N, K = [int(x) for x in os.environ.get("MNK", "32768,2560,2560").split(",")]
a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
epi = k.I2V_EPI_GEGLU if os.environ.get("EPI") == "geglu" else k.I2V_EPI_NONE
for _ in range(3):
    k.gemm(a, w, epilogue=epi)
torch.cuda.synchronize()
