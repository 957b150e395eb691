"""How far the fp32 reference itself is from exact arithmetic on the bench-size train steps of tests/test_hip_parity.py
(test_bench_size_train_step_matches_the_oracle): the oracle (oracle/qagnn_oracle.py, the CPU restatement pinned against the reference's
golden vectors) is run once in float32 and once in float64 on the same seeded weights and batch, and max|g32 - g64| is stored per
gradient tensor.  A candidate cannot be asked to sit closer to the fp32 run than the fp32 run sits to the exact answer: at the
64 x 4 = 256-subgraph OpenBookQA-shaped batch the two differ by a median 2.9e-2 of a tensor's scale (the ReLU kinks of ~50 M train-mode
BatchNorm outputs, see the test file), at the 320-subgraph CSQA batch by less.

Run from the repository root (CPU only, ~2 minutes and ~30 GB per workload):  python tests/golden/make_bench_size_yardstick.py
Writes tests/golden/bench_size_f64_yardstick.json.  Test infrastructure: nothing under qagnn_amd/ reads it.
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import helpers  # noqa: E402
import test_hip_parity as T  # noqa: E402
from oracle import qagnn_oracle as O  # noqa: E402


def yardstick(workload):
    ref = T._bench_size_case(workload)
    cfg, wl = ref['cfg'], ref['wl']
    nq, nc = wl['nq'], wl['nc']
    sv, cids, nt, ns, al, bei, bet = ref['inputs']
    torch.manual_seed(0)
    m = O.build_qagnn(cfg)
    helpers.det_fill_(m, 7, 0.6)
    m.pooler.dropout.p = m.pooler.attention.dropout.p = 0.0
    m = m.double().train()
    torch.set_default_dtype(torch.float64)
    O.PIN_FP32_SCORES = True
    try:
        lg, _ = m(sv.double(), cids, nt, ns.double(), al, (bei, bet))
        torch.nn.functional.cross_entropy(lg.view(nq, nc), ref['labels']).backward()
    finally:
        torch.set_default_dtype(torch.float32)
        O.PIN_FP32_SCORES = False
    out = {'__logits__': float((ref['logits'].double() - lg.detach()).abs().max())}
    for k, p in m.named_parameters():
        if p.grad is not None:
            out[k] = float(f'{(ref["grads"][k].double() - p.grad).abs().max().item():.4e}')
    T._BENCH_SIZE.pop(workload)  # (tens of GB of autograd state)
    return out


if __name__ == '__main__':
    res = {w: yardstick(w) for w in (sys.argv[1:] or list(T.BENCH_WORKLOADS))}
    path = os.path.join(HERE, 'bench_size_f64_yardstick.json')
    old = json.load(open(path)) if os.path.exists(path) else {}
    old.update(res)
    json.dump(old, open(path, 'w'), indent=0, sort_keys=True)
    print('wrote', path, {w: len(v) for w, v in old.items()})
