"""Which torch (aten) ops still launch work inside one UNet forward of the HIP model (they should be layout glue
only): profile one forward of the small test UNet on the GPU and list the aten ops with their Python call sites."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import parity
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
hu = parity.hip_unet_from_oracle(parity.oracle_small_unet(), dev)
inp = parity.small_unet_inputs()
args = (inp["sample"].to(dev), inp["timestep"].to(dev), True, inp["ctx"].to(dev))
with torch.no_grad():
    for _ in range(2):
        hu(*args)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        hu(*args)
        torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=3).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=50,
                                                  max_src_column_width=90))
