"""Attention against a torch fp32 reference as the logits grow (inputs scaled by `amp`): checks the deferred-max path."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
torch.manual_seed(0)
for d in (40, 80, 160):
    for lk in (64, 128, 512):
        for amp in (1.0, 2.0, 4.0):
            bq, hd, lq = 2, 8, 256; c = hd * d
            q = (torch.randn(bq * lq, c, device=dev) * amp).half(); kk = (torch.randn(bq * lk, c, device=dev) * amp).half()
            vt = torch.randn(bq, c, lk, device=dev).half()
            out = k.attention(q, kk, vt, batch_q=bq, lq=lq, lk=lk, heads=hd, head_dim=d, kv_group=1).view(bq, lq, hd, d).float()
            qq = q.view(bq, lq, hd, d).float().permute(0, 2, 1, 3); kb = kk.view(bq, lk, hd, d).float().permute(0, 2, 1, 3)
            vb = vt.view(bq, hd, d, lk).float()
            lg = qq @ kb.transpose(2, 3) * d ** -0.5
            ref = (torch.softmax(lg, -1) @ vb.transpose(2, 3)).permute(0, 2, 1, 3)
            err = (out - ref).abs()
            bad = (err > 0.02 * ref.abs().max()) | ~torch.isfinite(out)
            print(f"d={d} lk={lk} amp={amp}: max |logit| {lg.abs().max().item():6.1f}  err {err[torch.isfinite(err)].max().item() / ref.abs().max().item():.2e}"
                  f"  bad rows {int(bad.any(-1).sum())}/{bq * lq * hd}  nonfinite {int((~torch.isfinite(out)).sum())}", flush=True)
