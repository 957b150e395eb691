"""Time qagnn_gemm_tn_f32 on the weight-gradient shapes of a B = 320 step (run once per QAGNN_GEMM_SPLIT setting)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from qagnn_amd import ops  # noqa: E402

K = ops.kernels()
g = torch.Generator().manual_seed(0)
for R, Ka, No, aff in ((64000, 208, 208, False), (64000, 208, 208, True), (64000, 208, 624, False), (64000, 112, 624, False),
                       (2080, 612, 208, False)):
    A = torch.randn(R, Ka, generator=g).cuda()
    B = torch.randn(R, No, generator=g).cuda()
    kw = dict(a_scale=torch.randn(Ka, generator=g).cuda(), a_shift=torch.randn(Ka, generator=g).cuda()) if aff else {}
    out = torch.empty(Ka, No, device='cuda')
    for _ in range(5):
        K.gemm_tn(A, B, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        K.gemm_tn(A, B, out=out, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    if os.environ.get('TN_MICRO_NOREF'):
        print(f'R={R} Ka={Ka} No={No} affine={aff}: {us:8.1f} us  {2.0 * R * Ka * No / us / 1e6:7.1f} TFLOP/s')
        continue
    Ae = torch.relu(A * kw['a_scale'] + kw['a_shift']) if aff else A
    ref = Ae.double().t() @ B.double()
    err = ((out.double() - ref).abs() / (Ae.abs().double().t() @ B.abs().double()).clamp_min(1e-30)).max().item()
    print(f'GEMM_SPLIT={os.environ.get("QAGNN_GEMM_SPLIT", "1")} R={R} Ka={Ka} No={No} affine={aff}: {us:8.1f} us  '
          f'{2.0 * R * Ka * No / us / 1e6:7.1f} TFLOP/s  max err / sum|a||b| = {err:.2e}')
