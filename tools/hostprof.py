"""Steady-state host profile of the bench step (cProfile over steps AFTER warm-up): python tools/hostprof.py [questions] [steps]."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

q = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
wl = bench.WORKLOADS[bench.HEADLINE]
nc = wl['nc']
b = bench.to_device(bench.make_batch(wl, q, seed=123, n_concept=100000), dev, True, nc)
model = bench.build_model(bench.MQ, wl, 100000).to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
for _ in range(20):
    bench.step(model, b, nc, 1.0, params)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    bench.step(model, b, nc, 1.0, params)
torch.cuda.synchronize()
print(f'unprofiled: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step')
# host-only time: how long the Python side takes to ENQUEUE a step (no synchronisation inside)
t0 = time.perf_counter()
for _ in range(steps):
    bench.step(model, b, nc, 1.0, params)
t_host = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
print(f'host enqueue: {t_host * 1e3:.3f} ms/step')
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    bench.step(model, b, nc, 1.0, params)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(30)
st.sort_stats('cumulative').print_stats(70)
