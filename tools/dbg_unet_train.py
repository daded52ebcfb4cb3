import sys, os, torch
sys.path.insert(0, os.getcwd())
from tests.parity import hip_unet_from_oracle, host_threads, oracle_small_unet, small_unet_inputs
from i2v_adapter_unofficial_amd.training import UNetAdapterTrainer
dev = torch.device("cuda:0")
h = lambda t: t.half().float()
host_threads()
ou = oracle_small_unet(seed=77)
hu = hip_unet_from_oracle(ou, dev)
for prm in ou.parameters(): prm.requires_grad_(False)
train = {n: prm for n, prm in ou.named_parameters() if ".i2v_adapter.to_q." in n or ".i2v_adapter.to_out." in n}
for prm in train.values(): prm.requires_grad_(True)
inp = small_unet_inputs(b=2, f=4, hw=16)
t = torch.tensor([481, 481])
g = torch.Generator().manual_seed(78)
target = h(torch.randn(inp["sample"].shape, generator=g))
pred = ou(inp["sample"], t, True, inp["ctx"]).sample
mask = torch.ones_like(pred); mask[:, 0] = 0
loss = ((pred.float() - target) ** 2 * mask).sum() / mask.sum()
loss.backward()
tr = UNetAdapterTrainer(hu)
y = tr.forward(inp["sample"].half().to(dev), t.to(dev), inp["ctx"].half().to(dev))
for ls in (2.0 ** 12, 2.0 ** 4):
    if ls != 2.0 ** 12:
        y = tr.forward(inp["sample"].half().to(dev), t.to(dev), inp["ctx"].half().to(dev))
    got_loss, grads = tr.backward(target.to(dev), loss_scale=ls)
    print("loss scale", ls, "loss", got_loss.item(), loss.item())
    for name, prm in train.items():
        gg = grads[name].float().cpu(); r = prm.grad
        print(f"{name[:70]:70s} max|ref| {r.abs().max().item():.2e} max|got| {gg.abs().max().item():.2e} err/max {((gg - r).abs().max() / r.abs().max()).item():.2e}")
