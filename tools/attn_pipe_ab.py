"""head_dim-40 attention of the 64 x 64 level under I2V_ATTN_PIPE=0 / 1 (software-pipelined key loop): time and the max error
against a torch fp32 reference on two (batch) slices; self (kv_group 1) and cross-frame (kv_group 16) forms."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
tag = "PIPE=" + os.environ.get("I2V_ATTN_PIPE", "0")
torch.manual_seed(0)
for bq, grp, lq, lk, amp in [(32, 1, 4096, 4096, 1.0), (32, 16, 4096, 4096, 1.0), (16, 1, 4096, 4096, 3.0), (4, 1, 1024, 1024, 1.0),
                             (2, 1, 9216, 9216, 1.0), (4, 2, 1000, 960, 2.0)]:
    hd, d = 8, 40; c = hd * d
    q = (torch.randn(bq * lq, c, device=dev) * amp).half(); kk = (torch.randn(bq // grp * lk, c, device=dev) * amp).half()
    vt = torch.randn(bq // grp, c, lk, device=dev).half()
    run = lambda: k.attention(q, kk, vt, batch_q=bq, lq=lq, lk=lk, heads=hd, head_dim=d, kv_group=grp)
    out = run()
    # reference on the first and last batch entries
    err = 0.0
    for b in (0, bq - 1):
        qq = q.view(bq, lq, hd, d)[b].float().permute(1, 0, 2); kb = kk.view(bq // grp, lk, hd, d)[b // grp].float().permute(1, 0, 2)
        vb = vt.view(bq // grp, hd, d, lk)[b // grp].float()
        pr = torch.softmax(qq @ kb.transpose(1, 2) * d ** -0.5, dim=-1)
        ref = (pr @ vb.transpose(1, 2)).permute(1, 0, 2).reshape(lq, c)
        err = max(err, (out.view(bq, lq, c)[b].float() - ref).abs().max().item() / ref.abs().max().item())
    for _ in range(2): run()
    torch.cuda.synchronize(); n = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / n
    print(f"[{tag}] B{bq} grp{grp} Lq{lq} Lk{lk} amp{amp}: {ms * 1e3:8.1f} us {4.0 * bq * hd * lq * lk * d / ms / 1e9:7.1f} TFLOP/s  rel err {err:.2e}", flush=True)
