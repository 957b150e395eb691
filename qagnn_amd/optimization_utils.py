"""Optimisers behind the reference's OPTIMIZER_CLASSES interface (reference utils/optimization_utils.py:100-105).

`RAdam` keeps the reference's constructor, hyper-parameter groups, state layout (`step`, `exp_avg`, `exp_avg_sq` per
parameter, so optimizer.state_dict() round-trips with the reference's) and update rule (utils/optimization_utils.py:31-97),
but the update itself is ONE fused multi-tensor HIP launch sequence per (group, step count) instead of ~10 elementwise
kernels per parameter in a Python loop: the decoder has ~70 tensors / 2.85 M parameters, so a reference step is ~700 tiny
launches.  fp32 parameters on the GPU go through libqagnn_hip's qagnn_radam_step_f32; parameters that live on the CPU (unit
tests of the training driver) are updated with the same formulas in torch.
"""
import math

import torch
from torch.optim import SGD, Adam
from torch.optim import AdamW as _TorchAdamW
from torch.optim.optimizer import Optimizer


def radam_step_size(step, beta1, beta2, degenerated_to_sgd=True):
    """(N_sma, step_size) of the rectified update at `step` (utils/optimization_utils.py:63-80)."""
    beta2_t = beta2 ** step
    n_sma_max = 2 / (1 - beta2) - 1
    n_sma = n_sma_max - 2 * step * beta2_t / (1 - beta2_t)
    if n_sma >= 5:
        step_size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma * n_sma_max / (n_sma_max - 2)) / \
            (1 - beta1 ** step)
    elif degenerated_to_sgd:
        step_size = 1.0 / (1 - beta1 ** step)
    else:
        step_size = -1
    return n_sma, step_size


class AdamW(_TorchAdamW):
    """`transformers.AdamW` as the reference imports it (utils/optimization_utils.py:3; removed from current transformers): the
    decoupled-weight-decay update of torch.optim.AdamW with THAT class's defaults -- eps = 1e-6 and weight_decay = 0.0, not
    torch's 1e-8 / 0.01.  The reference's driver sets weight_decay per parameter group and never passes eps (qagnn.py:196-206),
    so `--optim adamw` trains with eps = 1e-6 there and must do so here."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, **kwargs):
        kwargs.pop('correct_bias', None)  # transformers' switch; True (its default) is what torch implements
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kwargs)


class RAdam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, degenerated_to_sgd=True):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(betas[1]))
        self.degenerated_to_sgd = degenerated_to_sgd
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group['betas']
            buckets = {}  # (step, on_gpu) -> lists: parameters that share a step count share step_size and the branch taken
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError('RAdam does not support sparse gradients')
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
                state['step'] += 1
                fused = p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
                buckets.setdefault((state['step'], fused, p.device if fused else None), []).append(p)
            for (step, fused, _dev), ps in buckets.items():
                n_sma, step_size = radam_step_size(step, beta1, beta2, self.degenerated_to_sgd)
                mode = 2 if n_sma >= 5 else (1 if step_size > 0 else 0)
                if fused:
                    from . import ops
                    grads = [p.grad if (p.grad.dtype == torch.float32 and p.grad.is_contiguous()) else p.grad.float().contiguous() for p in ps]
                    ops.kernels().radam_step(list(ps), grads, [self.state[p]['exp_avg'] for p in ps],
                                             [self.state[p]['exp_avg_sq'] for p in ps], beta1, beta2, group['eps'], group['lr'],
                                             group['weight_decay'], step_size, mode)
                    continue
                for p in ps:  # CPU tensors / other dtypes: the same update, tensor by tensor (utils/optimization_utils.py:52-93)
                    state = self.state[p]
                    grad = p.grad.float()
                    p32 = p.float()
                    m, v = state['exp_avg'], state['exp_avg_sq']
                    v.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
                    m.mul_(beta1).add_(grad, alpha=1 - beta1)
                    if mode == 0:
                        continue
                    if group['weight_decay'] != 0:
                        p32.add_(p32, alpha=-group['weight_decay'] * group['lr'])
                    if mode == 2:
                        p32.addcdiv_(m, v.sqrt().add_(group['eps']), value=-step_size * group['lr'])
                    else:
                        p32.add_(m, alpha=-step_size * group['lr'])
                    p.copy_(p32)
        return loss


OPTIMIZER_CLASSES = {
    'sgd': SGD,
    'adam': Adam,
    'adamw': AdamW,   # transformers.AdamW's defaults (eps 1e-6, weight_decay 0) over torch.optim.AdamW's update
    'radam': RAdam,
}
