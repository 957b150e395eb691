"""Time qagnn_gemm_nn (split route) on the product shapes of a B = 320 step, each alone on the GPU."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from qagnn_amd import ops  # noqa: E402

K = ops.kernels()
g = torch.Generator().manual_seed(0)
SMALL = '--small' in sys.argv   # the row counts of the host-bound configurations: 5 / 10 / 64 subgraphs
for M in ((500, 2000, 12800) if SMALL else (64000,)):
  print(f'== M = {M} rows, QAGNN_GEMM_SPLIT = {os.environ.get("QAGNN_GEMM_SPLIT", "1")}')
  ALL = (('mlp 208->208', 208, 0, 208, 208), ('proj [208|112]->624', 208, 112, 624, 208), ('dX 624->208', 624, 0, 208, 624),
         ('640->208', 640, 0, 208, 640), ('1024->208', 1024, 0, 208, 1024), ('416->208', 416, 0, 208, 416),
         ('dS 624->112', 624, 0, 112, 624), ('[dX|dS] 624->320', 624, 0, 320, 624))
  for name, K1, K2, No, lda in ([ALL[0], ALL[1], ALL[2], ALL[6]] if SMALL else ALL):  # (scripts/nn_micro_trace.py relies on this order)
      A1s = torch.randn(M, lda, generator=g).cuda()
      A1 = A1s[:, :K1]
      B1 = torch.randn(K1, No, generator=g).cuda()
      A2 = torch.randn(M, K2, generator=g).cuda() if K2 else None
      B2 = torch.randn(K2, No, generator=g).cuda() if K2 else None
      kw = dict(B1n=B1.t().contiguous(), B2n=B2.t().contiguous() if K2 else None)
      out = torch.empty(M, No, device='cuda')
      try:
          for _ in range(5):
              K.gemm_nn(A1, B1, A2, B2, out=out, **kw)
      except Exception as e:  # strided A is not accepted by every path
          print(f'{name}: {type(e).__name__} {e}')
          continue
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      n = 50
      e0.record()
      for _ in range(n):
          K.gemm_nn(A1, B1, A2, B2, out=out, **kw)
      e1.record()
      torch.cuda.synchronize()
      us = e0.elapsed_time(e1) * 1e3 / n
      nkt = -(-K1 // 32) + (-(-K2 // 32) if K2 else 0)
      ncb = -(-No // 208) if No >= 208 else 1
      print(f'{name:24s} {us:8.1f} us  {2.0 * M * (K1 + K2) * No / us / 1e6:7.1f} TFLOP/s   {us / (nkt * ncb):6.2f} us per (k-tile x column block)')
