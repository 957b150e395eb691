"""RAdam behind the reference's OPTIMIZER_CLASSES interface.

* the float64 oracle (oracle/radam_oracle.py) against golden vectors produced by the REFERENCE's own RAdam class;
* qagnn_amd.optimization_utils.RAdam on CPU tensors against the same vectors (state layout, parameter groups, lagging step counts);
* `-m gpu`: the fused multi-tensor HIP kernel (qagnn_radam_step_f32, through the C ABI) against the float64 oracle on the decoder's
  real tensor list, with gradients that are misaligned views of one flat buffer (what GatherPlan's backward hands out).
"""
import os
import sys

import numpy as np
import pytest
import torch

import helpers
sys.path.insert(0, os.path.join(helpers.ROOT, 'tests', 'golden'))
import make_golden_radam as G  # noqa: E402  (only its seeded tensor generators; main() needs /root/reference and is not called)
from oracle import radam_oracle as RO  # noqa: E402
from qagnn_amd import optimization_utils as OU  # noqa: E402

FIX = os.path.join(helpers.GOLDEN_DIR, 'radam.npz')


def _group_of(i):
    return next(g for g in G.GROUPS if i in g['idx'])


def test_oracle_matches_the_reference_optimizer():
    fix = np.load(FIX)
    p = [t.numpy().astype(np.float64) for t in G.tensors(7)]
    m = [np.zeros_like(x) for x in p]
    v = [np.zeros_like(x) for x in p]
    steps = [0] * len(p)
    for step in range(1, G.STEPS + 1):
        for i, g in enumerate(G.grads_at(step)):
            if step == 3 and i == 3:
                continue
            steps[i] += 1
            grp = _group_of(i)
            p[i], m[i], v[i] = RO.radam_step(p[i], g.numpy(), m[i], v[i], steps[i], grp['lr'], weight_decay=grp['weight_decay'])
            np.testing.assert_allclose(p[i], fix[f'p{i}_step{step}'], rtol=2e-6, atol=1e-7)
    for i in range(len(p)):
        assert steps[i] == int(fix[f'steps{i}'])
        np.testing.assert_allclose(m[i], fix[f'm{i}'], rtol=1e-5, atol=3e-7)
        np.testing.assert_allclose(v[i], fix[f'v{i}'], rtol=1e-5, atol=1e-8)


def _run_package(device):
    params = [torch.nn.Parameter(t.to(device)) for t in G.tensors(7)]
    opt = OU.OPTIMIZER_CLASSES['radam']([dict(params=[params[i] for i in g['idx']], lr=g['lr'], weight_decay=g['weight_decay'])
                                         for g in G.GROUPS], betas=(0.9, 0.999), eps=1e-8)
    hist = {}
    for step in range(1, G.STEPS + 1):
        for i, (p, g) in enumerate(zip(params, G.grads_at(step))):
            p.grad = None if (step == 3 and i == 3) else g.to(device)
        opt.step()
        for i, p in enumerate(params):
            hist[f'p{i}_step{step}'] = p.detach().cpu().numpy().copy()
    return params, opt, hist


def _check_against_fixture(params, opt, hist):
    fix = np.load(FIX)
    for k, val in hist.items():
        np.testing.assert_allclose(val, fix[k], rtol=3e-6, atol=2e-7, err_msg=k)
    for i, p in enumerate(params):
        st = opt.state[p]
        assert st['step'] == int(fix[f'steps{i}']) and set(st) == {'step', 'exp_avg', 'exp_avg_sq'}
        np.testing.assert_allclose(st['exp_avg'].cpu().numpy(), fix[f'm{i}'], rtol=1e-5, atol=3e-7)
        np.testing.assert_allclose(st['exp_avg_sq'].cpu().numpy(), fix[f'v{i}'], rtol=1e-5, atol=1e-8)
    sd = opt.state_dict()  # the reference's checkpoint layout (qagnn.py:322 saves optimizer state through state_dict())
    assert set(sd['param_groups'][0]) >= {'lr', 'betas', 'eps', 'weight_decay', 'params'}


def test_package_radam_cpu_matches_the_reference_optimizer():
    _check_against_fixture(*_run_package('cpu'))


def test_optimizer_classes_interface():
    assert set(OU.OPTIMIZER_CLASSES) == {'sgd', 'adam', 'adamw', 'radam'}
    for bad in (dict(lr=-1.0), dict(eps=-1.0), dict(betas=(1.0, 0.9)), dict(betas=(0.9, 1.0))):
        with pytest.raises(ValueError):
            OU.RAdam([torch.nn.Parameter(torch.zeros(2))], **bad)
    p = torch.nn.Parameter(torch.ones(3))
    opt = OU.RAdam([p], lr=0.1, degenerated_to_sgd=False)
    p.grad = torch.ones(3)
    opt.step()   # N_sma < 5 and no SGD fallback: moments move, the parameter does not (:76-77)
    assert torch.equal(p.detach(), torch.ones(3)) and float(opt.state[p]['exp_avg'][0]) > 0


@pytest.mark.gpu
def test_fused_radam_matches_the_reference_vectors_on_gpu():
    from qagnn_amd import ops
    ops.set_kernels(None)
    _check_against_fixture(*_run_package('cuda'))


@pytest.mark.gpu
@pytest.mark.parametrize('step,wd', [(1, 0.01), (4, 0.0), (6, 0.01), (500, 0.01)])
def test_fused_radam_decoder_tensor_list_vs_float64(step, wd):
    """All trainable tensors of the CSQA decoder (74 tensors, 2.85 M parameters: several kernel-argument packs, tensors split
    over pack boundaries), gradients as misaligned views into one flat buffer, one step at a given step count."""
    from qagnn_amd import modeling_qagnn as MQ
    from qagnn_amd import ops
    ops.set_kernels(None)
    torch.manual_seed(0)
    model = MQ.QAGNN(None, 5, 4, 38, 1024, 3000, 200, 1024, 2, 200, 0, 0.2, 0.2, 0.2).cuda()
    params = [p for p in model.parameters() if p.requires_grad]
    total = sum(p.numel() for p in params)
    assert len(params) >= 70 and total > 2_500_000
    g = torch.Generator(device='cuda').manual_seed(step)
    flat = torch.randn(total + 3, generator=g, device='cuda')[3:]      # 12-byte offset: no view is 16-byte aligned by construction
    off, p0 = 0, []
    for p in params:
        p.grad = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
        p0.append(p.detach().cpu().numpy().astype(np.float64))
    opt = OU.RAdam(params, lr=1e-3, weight_decay=wd)
    m0 = [0.01 * torch.randn_like(p) for p in params]
    v0 = [(0.01 * torch.randn_like(p)) ** 2 for p in params]
    for p, m, v in zip(params, m0, v0):
        opt.state[p] = dict(step=step - 1, exp_avg=m.clone(), exp_avg_sq=v.clone())
    opt.step()
    torch.cuda.synchronize()
    for p, pb, m, v in zip(params, p0, m0, v0):
        rp, rm, rv = RO.radam_step(pb, p.grad.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy(), step, 1e-3, weight_decay=wd)
        st = opt.state[p]
        assert st['step'] == step
        np.testing.assert_allclose(st['exp_avg'].cpu().numpy(), rm, rtol=2e-6, atol=3e-8)   # two O(0.1) terms may cancel
        np.testing.assert_allclose(st['exp_avg_sq'].cpu().numpy(), rv, rtol=2e-6, atol=1e-10)
        np.testing.assert_allclose(p.detach().cpu().numpy(), rp, rtol=2e-6, atol=1e-7)


def test_adamw_keeps_the_defaults_and_the_update_rule_of_the_class_the_reference_imports():
    """The reference's OPTIMIZER_CLASSES['adamw'] is transformers.AdamW (utils/optimization_utils.py:3, 103; transformers 3.4.0): eps
    1e-6, weight_decay 0, and an update rule that differs from torch.optim.AdamW's (eps added to the uncorrected sqrt(v), decay after the
    Adam update).  Its driver never passes eps (qagnn.py:196-206).  Held to oracle.radam_oracle.transformers_adamw_step in float64."""
    import numpy as np
    import torch
    from oracle import radam_oracle as RO
    p = torch.nn.Parameter(torch.ones(3))
    opt = OU.OPTIMIZER_CLASSES['adamw']([{'params': [p], 'weight_decay': 0.01, 'lr': 1e-3}])
    g = opt.param_groups[0]
    assert g['eps'] == 1e-6 and g['weight_decay'] == 0.01 and g['betas'] == (0.9, 0.999) and g['correct_bias'] is True
    assert OU.OPTIMIZER_CLASSES['adamw']([p], lr=1e-3).param_groups[0]['weight_decay'] == 0.0
    p.grad = torch.full((3,), 0.5)
    opt.step()
    # first step by hand: m = 0.05, sqrt(v) = 0.5 sqrt(0.001); p -= lr sqrt(0.001) / 0.1 * m / (sqrt(v) + eps); then p -= lr wd p
    upd = 1e-3 * (0.001 ** 0.5) / 0.1 * 0.05 / (0.5 * 0.001 ** 0.5 + 1e-6)
    want = (1.0 - upd) * (1.0 - 1e-3 * 0.01)
    assert torch.allclose(p.detach(), torch.full((3,), want), rtol=0, atol=1e-7)
    torch_first = 1.0 - 1e-3 * 0.01 - 1e-3 * 0.5 / (0.5 + 1e-6)   # what torch.optim.AdamW(eps=1e-6) does on the same step
    assert abs(want - torch_first) > 5e-8  # the two rules are not the same rule (eps / sqrt(1 - b2^t) ~ 3e-5 here)
    # several steps, two groups with different decay, against the float64 restatement
    gen = torch.Generator().manual_seed(5)
    ps = [torch.nn.Parameter(torch.randn(7, 5, generator=gen)), torch.nn.Parameter(torch.randn(11, generator=gen))]
    opt = OU.AdamW([{'params': [ps[0]], 'weight_decay': 0.01}, {'params': [ps[1]], 'weight_decay': 0.0}], lr=2e-3)
    ref = [(q.detach().double().numpy().copy(), np.zeros(q.shape), np.zeros(q.shape)) for q in ps]
    for step in range(1, 8):
        for q in ps:
            q.grad = torch.randn(q.shape, generator=gen) * (0.1 if step % 2 else 1e-4)  # small gradients: where eps placement matters
        opt.step()
        ref = [RO.transformers_adamw_step(rp, q.grad.double().numpy(), rm, rv, step, 2e-3, weight_decay=wd)
               for (rp, rm, rv), q, wd in zip(ref, ps, (0.01, 0.0))]
        for q, (rp, rm, rv) in zip(ps, ref):
            st = opt.state[q]
            assert st['step'] == step
            np.testing.assert_allclose(st['exp_avg'].numpy(), rm, rtol=3e-6, atol=1e-9)
            np.testing.assert_allclose(st['exp_avg_sq'].numpy(), rv, rtol=3e-6, atol=1e-12)
            np.testing.assert_allclose(q.detach().numpy(), rp, rtol=3e-6, atol=2e-7)
