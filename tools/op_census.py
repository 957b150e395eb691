"""Count the aten ops / kernels of one bench step (torch.profiler), to see where the small-kernel launches come from."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from qagnn_amd import modeling_qagnn as MQ  # noqa: E402

dev = torch.device('cuda', 0)
b = {k: v.to(dev) for k, v in bench.make_batch(64, seed=1000, n_concept=100000).items()}
model = bench.build_model(MQ, 100000, p=0.2).to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
for _ in range(3):
    bench.step(model, b, 1, params)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
    bench.step(model, b, 1, params)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cpu_time_total', row_limit=60, max_name_column_width=60))
